#!/bin/bash
# builds and runs the conv micro-benchmark on the GPU box: tools/conv_bench.sh [debug masks...]
set -e
cd "$(dirname "$0")/.."
# production kernel (debug mask 0) and, with -DUS_CONV_ABLATE, the timing-ablation build for masks != 0
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench
if [ "$1" = f16 ]; then
  # f16x3 GEMM forms on the Winograd-domain shapes (production build only): tools/conv_bench.sh f16 [shape filter]
  /tmp/conv_bench 0 | grep "calibration"
  CB_ONLY="${2:-G}" CB_F16=1 /tmp/conv_bench 9
  CB_ONLY="${2:-G}" /tmp/conv_bench 9
  exit 0
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUS_CONV_ABLATE tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench_ablate
for pm in 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUS_PRIO_MODE=$pm tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/conv_bench_prio$pm
done
if [ "$1" = prio ]; then
  for r in 1 2; do for pm in 0 1 2; do echo "== PRIO_MODE $pm (round $r)"; if [ $pm = 0 ]; then /tmp/conv_bench 9 | grep "tm= 64"; else /tmp/conv_bench_prio$pm 9 | grep "tm= 64"; fi; done; done
  exit 0
fi
for m in "${@:-0}"; do if [ "$m" = 0 ]; then /tmp/conv_bench 0; else /tmp/conv_bench_ablate $m; fi; done
