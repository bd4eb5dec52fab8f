set -u
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench || exit 1
for f in "F1 fused 256" "F1 fused 512" "F2" "F3" "F0"; do
  CB_ONLY="$f" CB_F16=1 CB_AB=128 CB_TM=64 CB_STATS=1 CB_COLD=1 /tmp/conv_bench 9
  CB_ONLY="$f" CB_F16=1 CB_AB=128 CB_AB_FIRST=1 CB_TM=64 CB_STATS=1 CB_COLD=1 /tmp/conv_bench 9
done
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-120
python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-120
