#!/bin/bash
# Counter evidence for profiles/: separate rocprofv3 --pmc passes (never combined with trace domains other than --kernel-trace)
# over one short bench.py sampling call, then tools/pmc_summary.py joins them per kernel instantiation and per U-Net launch.
# usage (GPU box, repo root): [PMC_BENCH_ARGS='--batch 8'] tools/pmc_collect.sh <tag>
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
pass() {   # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o p -- python3 bench.py --steps 1 --warmup 0 --diffusion-steps 4 --no-cpu-baseline --no-anchor ${PMC_BENCH_ARGS:-} > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "pass $name produced no counters"; tail -3 "$out/$name.log"; return 1; fi
  mv "$f" "$out/$name.csv"; rm -rf "$out/$name"
  echo "pass $name ok ($(wc -l < "$out/$name.csv") rows)"
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
pass mfma SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES || echo "(mfma pass skipped)"
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass grbm GRBM_GUI_ACTIVE || echo "(grbm pass skipped)"
python3 tools/pmc_summary.py "$out" "$out/summary.json" | tail -40
