set -e
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/cb
for r in 1 2; do
echo "== in-kernel split"; CB_ONLY="1x1" CB_F16=1 /tmp/cb 9 | grep TFLOP
echo "== pre-split A"; CB_ONLY="1x1" CB_F16=1 CB_PRESPLIT=1 /tmp/cb 9 | grep TFLOP
done
