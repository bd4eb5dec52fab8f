#!/bin/bash
# A/B of environment switches on the GPU box, interleaved: tools/exp_ab.sh "<env A>" "<env B>" [rounds]   (empty string = defaults)
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --profile-steps 1"
n=${3:-2}
for r in $(seq 1 $n); do
  env $1 $B > gpurun_out/ab_A$r.log 2>&1
  env $2 $B > gpurun_out/ab_B$r.log 2>&1
done
python tools/bench_line.py gpurun_out/ab_A*.log gpurun_out/ab_B*.log
