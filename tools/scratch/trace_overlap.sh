#!/bin/bash
# kernel trace of the eager fine-tune loop; prints sum of kernel durations, union of busy intervals and wall span per iteration
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_finetune.py --iters 10 --warmup 2 --no-cpu-baseline --no-graph "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
# last 60 % of the run = steady state
t0 = ev[0][0]; t1 = ev[-1][1]
lo = t0 + (t1 - t0) * 4 // 10
ev = [e for e in ev if e[0] >= lo]
span = ev[-1][1] - ev[0][0]
ssum = sum(e[1] - e[0] for e in ev)
union = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _, _ in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
qs = {}
for s, e, n, q in ev:
    qs.setdefault(q, [0, 0]); qs[q][0] += 1; qs[q][1] += e - s
print(f"span {span/1e6:.2f} ms  sum of durations {ssum/1e6:.2f} ms  union busy {union/1e6:.2f} ms  idle {(span-union)/1e6:.2f} ms  overlap {(ssum-union)/1e6:.2f} ms")
for q, (n, d) in sorted(qs.items()): print(f"  queue {q}: {n} kernels, {d/1e6:.2f} ms")
PY
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-120
