#!/bin/bash
# samples clocks / power with rocm-smi while bench.py runs (B from $1, default 1)
B=${1:-1}
python bench.py --steps 40 --warmup 2 --no-cpu-baseline --batch $B > gpurun_out/ps_bench.log 2>&1 &
pid=$!
sleep 13
for i in $(seq 1 14); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' '|'
  echo
  sleep 0.5
done
wait $pid
tail -1 gpurun_out/ps_bench.log | cut -c1-200
