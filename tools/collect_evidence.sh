#!/bin/bash
# Round evidence on the GPU box in one call: tools/collect_evidence.sh <round tag, e.g. r03>
# Writes gpurun_out/<tag>_evidence/: bench lines (B=1 default invocation, B=8, B=64, fine-tune, pre-training, front end, 2-rank rehearsal),
# rocprofv3 kernel stats + per-launch conv table, PMC summary.  Copy what should be judged into profiles/.
tag=${1:-r04}
out=gpurun_out/${tag}_evidence
rm -rf "$out"; mkdir -p "$out"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_B1.json" 2> "$out/bench_B1.err"; echo "B1 rc=$?"
python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_B8.json" 2>/dev/null; echo "B8 rc=$?"
python bench.py --config 64x1 --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_B64.json" 2>/dev/null; echo "B64 rc=$?"
python bench_finetune.py --iters 100 --check 3 > "$out/bench_finetune.json" 2>/dev/null; echo "finetune rc=$?"
python bench_pretrain.py --iters 10 > "$out/bench_pretrain.json" 2>/dev/null; echo "pretrain rc=$?"
python bench_frontend.py > "$out/bench_frontend.json" 2>/dev/null; echo "frontend rc=$?"
python bench.py --gpus 2 --rehearse-on-one-gpu --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_rehearsal_2ranks.json" 2>/dev/null; echo "rehearsal rc=$?"
tools/profile_bench.sh ${tag}_evidence/prof > "$out/profile.log" 2>&1; echo "profile rc=$?"
tools/profile_finetune.sh ${tag}_evidence/ftprof > "$out/profile_finetune.log" 2>&1; echo "ft profile rc=$?"
tools/pmc_collect.sh ${tag}_evidence/pmc > "$out/pmc.log" 2>&1; echo "pmc rc=$?"
# configs[2]'s launches (B' = 24 per micro-batch of 8): per-launch table and counters
AT_ARGS="--bp 24" tools/profile_bench.sh ${tag}_evidence/prof_B8 --batch 8 > "$out/profile_B8.log" 2>&1; echo "profile B8 rc=$?"
PMC_BENCH_ARGS="--batch 8" tools/pmc_collect.sh ${tag}_evidence/pmc_B8 > "$out/pmc_B8.log" 2>&1; echo "pmc B8 rc=$?"
for f in bench_B1 bench_B8 bench_B64 bench_rehearsal_2ranks; do python tools/bench_line.py "$out/$f.json"; done
tail -1 "$out/bench_finetune.json" | cut -c1-160
tail -1 "$out/bench_pretrain.json" | cut -c1-200
tail -3 "$out/pmc.log"
