#!/bin/bash
# builds and runs the conv micro-benchmark on the GPU box: tools/conv_bench.sh [debug masks...] | f16 [filter] | diag [filter] | prio
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
if [ "$1" = life ]; then
  # workgroup lifetimes of the production tile choice (100 MHz stamps; -DUS_LIFE): tools/conv_bench.sh life [shape filter]
  # (US_LIFE_DRAIN: the exit stamp waits for the wave's stores; without it the wave ends with its stores in flight, as in production)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DUS_LIFE ${LIFE_FLAGS:-} tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip -o /tmp/cb_life
  for tm in ${LIFE_TMS:-0 256 128}; do echo "== CB_TM=$tm"; CB_ONLY="${2:-H3}" CB_F16=1 CB_TM=$tm /tmp/cb_life 9 | grep "TFLOP\|life\|times"; done
  exit 0
fi
if [ "$1" = diag ]; then
  # where the three-buffer f16x3 GEMM spends a step: s_memtime stamps, then the timing ablations (results of the
  # ablated builds are wrong by construction): tools/conv_bench.sh diag [shape filter]
  B="hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp unitspeech_amd/csrc/conv_igemm.hip unitspeech_amd/csrc/ops.hip"
  $B -DUS_STAMP -o /tmp/cb_stamp & $B -DUS_ABL_LDS -o /tmp/cb_lds & $B -DUS_ABL_DMA -o /tmp/cb_dma & $B -DUS_ABL_DMA -DUS_ABL_LDS -o /tmp/cb_both & $B -DUS_DMA_SPREAD=0 -o /tmp/cb_burst & wait
  for v in "" _stamp _lds _dma _both _burst; do
    echo "== build: ${v:-production} (stamp: cycles per step; lds: half the fragment reads; dma: no loads in the loop; burst: LDS-DMA pieces back to back)"
    if [ -z "$v" ]; then CB_ONLY="${2:-G3 gemm 1024}" CB_F16=1 CB_TM=256 /tmp/conv_bench 9 | grep "TFLOP\|stamps"; else CB_ONLY="${2:-G3 gemm 1024}" CB_F16=1 CB_TM=256 /tmp/cb$v 9 | grep "TFLOP\|stamps"; fi
  done
  echo "== whole-frequency XCD placement off / on"
  CB_NO_XCDZ=1 CB_ONLY="${2:-G}" CB_F16=1 CB_TM=256 /tmp/conv_bench 9 | grep TFLOP
  CB_ONLY="${2:-G}" CB_F16=1 CB_TM=256 /tmp/conv_bench 9 | grep TFLOP
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
