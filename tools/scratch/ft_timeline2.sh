#!/bin/bash
# kernel trace of the default fine-tune loop, one steady-state iteration written out launch by launch (queue, start, duration, gap to the
# previous launch of the same queue): gpurun_out/ft_timeline/iter.txt + a per-queue / per-kernel digest
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ft_timeline
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_finetune.py --iters 30 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" "$out/iter.txt" <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
packs = [i for i, e in enumerate(ev) if "pack_table_kernel" in e[2]]
a, b = packs[-3], packs[-2]
seg = ev[a:b]
t0 = seg[0][0]
last = {}
with open(sys.argv[2], "w") as f:
    for s, e, n, q in seg:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        nm = n.replace("us::", "").replace("(anonymous namespace)::", "")[:70]
        f.write(f"{(s - t0) / 1e3:9.1f} q{q:>3} dur {(e - s) / 1e3:7.1f} gap {gap:7.1f}  {nm}\n")
qs = collections.defaultdict(list)
for s, e, n, q in seg:
    qs[q].append((s, e, n))
print(f"iteration: {(ev[b][0] - t0) / 1e6:.2f} ms, {len(seg)} launches")
for q, l in qs.items():
    busy = sum(e - s for s, e, _ in l)
    gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
    print(f"queue {q}: {len(l)} launches, busy {busy / 1e6:.2f} ms, span {(l[0][0] - t0) / 1e6:.2f}..{(l[-1][1] - t0) / 1e6:.2f} ms, "
          f"gaps: sum {sum(g for g in gaps if g > 0) / 1e6:.2f} ms, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, >20us: {sum(1 for g in gaps if g > 20000)}")
PY
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-120
