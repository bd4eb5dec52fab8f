set -u
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench || exit 1
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
for f in "G3 gemm 1024" "G2 gemm 512" "G1 gemm 256"; do
  CB_ONLY="$f" CB_F16=1 CB_AB=512 CB_TM=256 CB_COLD=1 /tmp/conv_bench 9
  CB_ONLY="$f" CB_F16=1 CB_AB=512 CB_AB_FIRST=1 CB_TM=256 CB_COLD=1 /tmp/conv_bench 9
done
for f in "L0 3x3" "L0 1x1 128->384" "L2 1x1 512"; do
  CB_ONLY="$f" CB_F16=1 CB_AB=512 CB_TM=64 CB_COLD=1 /tmp/conv_bench 9
  CB_ONLY="$f" CB_F16=1 CB_AB=512 CB_AB_FIRST=1 CB_TM=64 CB_COLD=1 /tmp/conv_bench 9
done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-120
US_F16_M16=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-120
