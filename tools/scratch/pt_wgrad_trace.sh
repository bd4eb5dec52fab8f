#!/bin/bash
# per-launch durations of the weight-gradient kernels of one pre-training step (32 crops), in launch order
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pt_trace
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_pretrain.py --iters 4 --warmup 2 "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Grid_Size", ""), r.get("Workgroup_Size","")) for r in rows)
packs = [i for i, e in enumerate(ev) if "pack_table_kernel" in e[2]]
a, b = packs[-2], packs[-1]
seg = ev[a:b]
t0 = seg[0][0]
print(f"step {(ev[b][0]-t0)/1e6:.2f} ms, {len(seg)} launches")
qs = collections.defaultdict(list)
for s, e, n, q, g, w in seg: qs[q].append((s, e, n))
for q, l in qs.items():
    print(f"queue {q}: {len(l)} launches busy {sum(e-s for s,e,_ in l)/1e6:.2f} ms span {(l[0][0]-t0)/1e6:.2f}..{(l[-1][1]-t0)/1e6:.2f}")
import os
pat = os.environ.get("PT_KERNELS", "wgrad_f16,wgrad_lds").split(",")
for s, e, n, q, g, w in seg:
    if any(p in n for p in pat):
        print(f"  {(s-t0)/1e3:9.1f} q{q} {(e-s)/1e3:8.1f} us  {n.replace('us::','')[:60]}")
PY
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-200
