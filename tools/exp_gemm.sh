#!/bin/bash
# micro-benchmark A/B of one runtime debug bit of the three-buffer f16x3 GEMM: tools/exp_gemm.sh <bit> [filter] [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 ${3:-} tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench
for r in 1 2; do
  CB_AB=$1 CB_ONLY="${2:-G}" CB_F16=1 CB_TM=256 /tmp/conv_bench 9 | grep TFLOP
done
