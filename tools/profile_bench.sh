#!/bin/bash
# rocprofv3 kernel trace of one bench.py sampling call -> per-kernel stats + per-launch conv table under gpurun_out/<tag>/
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [extra bench.py args]; environment switches pass through
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 bench.py --steps 1 --warmup 0 --profile-steps 1 --no-cpu-baseline --no-anchor "$@" > "$out/bench.log" 2>&1
trace=$(find "$out" -name '*kernel_trace.csv' | head -1)
stats=$(find "$out" -name '*kernel_stats.csv' | head -1)
if [ -z "$trace" ] || [ -z "$stats" ]; then echo "no trace produced"; tail -5 "$out/bench.log"; exit 1; fi
python3 tools/analyze_trace.py "$trace" ${AT_ARGS:-} > "$out/conv_per_launch.txt" || true
cp "$stats" "$out/kernel_stats.csv"
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/conv_per_launch.txt"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
n_eval = max(sum(int(r["Calls"]) for r in rows if "final_conv_kernel" in r["Name"]) // 1, 1)      # one final_conv launch per evaluation
print(f"GPU time per evaluation ({n_eval} evaluations + set-up in the trace): {tot / n_eval / 1e6:.3f} ms")
for r in rows[:16]:
    print(f"  {int(r['TotalDurationNs']) / n_eval / 1e3:8.1f} us/eval  {int(r['Calls']):5d} calls  {r['Name'][:100]}")
PY
