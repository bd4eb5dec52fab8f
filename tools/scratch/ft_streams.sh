#!/bin/bash
# per-stream view of the eager backward of a fine-tune iteration: for each HIP queue, busy time and span between final_bwd_kernel and sumsq_kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ft_streams
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_finetune.py --iters 30 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
fb = [i for i, e in enumerate(ev) if "final_bwd_kernel" in e[2]]
ss = [i for i, e in enumerate(ev) if "sumsq_kernel" in e[2]]
fb, ss = fb[len(fb) // 2:], ss[len(ss) // 2:]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])       # queue -> busy, span, launches, last end - bwd start
n = 0
names = collections.defaultdict(lambda: collections.Counter())
for a in fb:
    b = next((j for j in ss if j > a), None)
    if b is None: continue
    seg = ev[a:b]
    t0 = seg[0][0]
    n += 1
    perq = collections.defaultdict(list)
    for s, e, k, q in seg: perq[q].append((s, e, k))
    for q, l in perq.items():
        acc[q][0] += sum(e - s for s, e, _ in l); acc[q][1] += l[-1][1] - l[0][0]; acc[q][2] += len(l); acc[q][3] += l[-1][1] - t0
        for s, e, k in l: names[q][k.split("(")[0][:50]] += e - s
    acc["_total"][1] += seg[-1][1] - t0
for q, v in acc.items():
    print(f"queue {q}: busy {v[0]/n/1e6:.2f} ms, first..last {v[1]/n/1e6:.2f} ms, {v[2]/n:.0f} launches, last end at {v[3]/n/1e6:.2f} ms after the backward's start")
    for k, t in names[q].most_common(6): print(f"      {t/n/1e3:7.1f} us  {k}")
PY
find "$out" -name '*kernel_trace.csv' -delete
