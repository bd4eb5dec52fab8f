set -e
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 1"
$B > gpurun_out/x_base.log 2>&1
US_TM_1X1=128 $B > gpurun_out/x_1x1_128.log 2>&1
US_TM_1X1=256 $B > gpurun_out/x_1x1_256.log 2>&1
US_TM_TAPS=128 $B > gpurun_out/x_taps_128.log 2>&1
US_TM_TAPS=256 $B > gpurun_out/x_taps_256.log 2>&1
US_TM_PRESPLIT=128 $B > gpurun_out/x_pre_128.log 2>&1
US_TM_PRESPLIT=256 $B > gpurun_out/x_pre_256.log 2>&1
$B > gpurun_out/x_base2.log 2>&1
python tools/bench_line.py gpurun_out/x_*.log
