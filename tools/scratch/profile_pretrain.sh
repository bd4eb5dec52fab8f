#!/bin/bash
# rocprofv3 kernel stats of the pre-training step (B=32 crops): tools/scratch/profile_pretrain.sh <tag>
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
ITERS=6
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 bench_pretrain.py --iters $ITERS --warmup 2 "$@" > "$out/bench.log" 2>&1
stats=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$stats" "$out/kernel_stats.csv"
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-330
python3 - "$out/kernel_stats.csv" $ITERS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) + 2
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel durations per step: {tot / n / 1e6:.2f} ms")
for r in rows[:22]:
    print(f"  {int(r['TotalDurationNs']) / n / 1e3:9.1f} us/step {int(r['Calls']) / n:7.1f} calls/step  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
