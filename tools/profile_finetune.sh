#!/bin/bash
# rocprofv3 kernel stats of the fine-tune bench (eager launches so that every kernel shows): tools/profile_finetune.sh <tag> [bench_finetune args]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
ITERS=20
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 bench_finetune.py --iters $ITERS --warmup 2 --no-cpu-baseline --no-graph "$@" > "$out/bench.log" 2>&1
stats=$(find "$out" -name '*kernel_stats.csv' | head -1)
if [ -z "$stats" ]; then echo "no stats produced"; tail -5 "$out/bench.log"; exit 1; fi
cp "$stats" "$out/kernel_stats.csv"
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-160
python3 - "$out/kernel_stats.csv" $ITERS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) + 2
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"GPU time per iteration ({n} iterations incl. warm-up + set-up in the trace): {tot / n / 1e6:.3f} ms, {calls / n:.0f} launches")
for r in rows[:28]:
    print(f"  {int(r['TotalDurationNs']) / n / 1e3:8.1f} us/iter  {int(r['Calls']) / n:6.1f} calls/iter  {r['Name'][:110]}")
PY
