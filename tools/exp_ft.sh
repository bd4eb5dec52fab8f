#!/bin/bash
# fine-tune bench variants, interleaved over rounds: tools/exp_ft.sh rounds "<env 1>" "<env 2>" ...  ("-" = defaults; FT_ARGS = extra bench arguments)
B="python bench_finetune.py --iters 100 --warmup 5 --no-cpu-baseline ${FT_ARGS:-}"
n=$1; shift
for r in $(seq 1 $n); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then $B > gpurun_out/ft_${i}_r$r.log 2>&1; else env $v $B > gpurun_out/ft_${i}_r$r.log 2>&1; fi
  done
done
i=0
for v in "$@"; do
  i=$((i+1)); echo "== variant $i: $v"
  for f in gpurun_out/ft_${i}_r*.log; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]}: {d['value']*1e3:.3f} ms/iter  host {d.get('host_enqueue_s_per_iter', 0)*1e3:.3f} ms  launch={d['config'].get('launch')}  last_loss={d['last_loss']:.6f}")
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
  done
done
