#!/bin/bash
# pre-training step variants, interleaved: tools/scratch/exp_pt.sh rounds "<env 1>" "<env 2>" ...  ("-" = defaults)
B="python bench_pretrain.py --iters 8 --warmup 2"
n=$1; shift
for r in $(seq 1 $n); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then $B > gpurun_out/pt_${i}_r$r.log 2>&1; else env $v $B > gpurun_out/pt_${i}_r$r.log 2>&1; fi
  done
done
i=0
for v in "$@"; do
  i=$((i+1)); echo "== variant $i: $v"
  for f in gpurun_out/pt_${i}_r*.log; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"  {d['ms_per_step']:.2f} ms/step {d['value']:.1f} crops/s  {d['ms_breakdown']}  loss {d['last_loss']:.7f}\")" 2>/dev/null || tail -2 $f; done
done
