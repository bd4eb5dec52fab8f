set -u
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench || exit 1
for f in "L0 3x3" "F0" "F1 fused 256"; do CB_ONLY="$f" CB_F16=1 CB_TM=64 /tmp/conv_bench 9;  CB_ONLY="$f" CB_F16=1 CB_TM=64 CB_STATS=1 /tmp/conv_bench 9; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests_4.log 2>&1; tail -4 gpurun_out/r2_gpu_tests_4.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_f16_d.json 2> gpurun_out/r2_bench_f16_d.err; cut -c1-140 gpurun_out/r2_bench_f16_d.json
US_WINO_MIN_LEVEL=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_f16_d_w1.json 2>/dev/null; cut -c1-140 gpurun_out/r2_bench_f16_d_w1.json
