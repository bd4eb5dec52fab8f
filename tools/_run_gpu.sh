set -u
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench || exit 1
for f in "G3 gemm 1024" "G2 gemm 512" "G1 gemm 256" "L0 3x3" "L2 1x1 512" "L0 1x1 128->384"; do
  CB_ONLY="$f" CB_F16=1 CB_AB=256 CB_COLD=1 /tmp/conv_bench 9 | grep -v "tm=128"
  CB_ONLY="$f" CB_F16=1 CB_AB=256 CB_AB_FIRST=1 CB_COLD=1 /tmp/conv_bench 9 | grep -v "tm=128"
done
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_f16_e.json 2> gpurun_out/r2_bench_f16_e.err; cut -c1-140 gpurun_out/r2_bench_f16_e.json
