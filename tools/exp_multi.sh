#!/bin/bash
# several environment variants, interleaved over rounds: tools/exp_multi.sh rounds "<env 1>" "<env 2>" ...  ("-" = defaults)
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 1"
n=$1; shift
for r in $(seq 1 $n); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then $B > gpurun_out/mv_${i}_r$r.log 2>&1; else env $v $B > gpurun_out/mv_${i}_r$r.log 2>&1; fi
  done
done
i=0
for v in "$@"; do i=$((i+1)); echo "== variant $i: $v"; python tools/bench_line.py gpurun_out/mv_${i}_r*.log; done
