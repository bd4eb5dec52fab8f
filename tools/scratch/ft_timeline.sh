#!/bin/bash
# kernel trace of the default fine-tune loop: where the iteration's wall time goes (phases delimited by pack_table / final_bwd / sumsq / adam)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ft_timeline
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_finetune.py --iters 30 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
packs = [i for i, e in enumerate(ev) if "pack_table_kernel" in e[2]]
packs = packs[len(packs) // 2:]                      # steady state
its = []
for a, b in zip(packs[:-1], packs[1:]):
    seg = ev[a:b]
    t0 = seg[0][0]
    def first(name):
        return next((e for e in seg if name in e[2]), None)
    fb = first("final_bwd_kernel"); ss = first("sumsq_kernel"); ad = first("adam_kernel")
    if not (fb and ss and ad):
        continue
    fwd_end = max(e[1] for e in seg if e[0] < fb[0])
    bwd_end = max(e[1] for e in seg if e[0] < ss[0])
    busy = 0; cs, ce = seg[0][0], seg[0][1]
    for s, e, _, _ in seg[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    its.append(((b and ev[b][0]) - t0, fwd_end - t0, fb[0] - fwd_end, bwd_end - fb[0], ss[0] - bwd_end, ad[1] - ss[0], ev[b][0] - ad[1], busy, sum(e[1] - e[0] for e in seg), len(seg)))
n = len(its)
avg = [sum(x[i] for x in its) / n / 1e6 for i in range(9)] + [sum(x[9] for x in its) / n]
print(f"{n} iterations: period {avg[0]:.2f} ms = forward (pack..objective) {avg[1]:.2f} + gap {avg[2]:.2f} + backward {avg[3]:.2f} + gap {avg[4]:.2f} + clip/Adam {avg[5]:.2f} + gap to next pack {avg[6]:.2f}")
print(f"   union busy {avg[7]:.2f} ms, sum of kernel durations {avg[8]:.2f} ms, {avg[9]:.0f} kernels")
PY
find "$out" -name '*kernel_trace.csv' -delete
tail -1 "$out/bench.log" | cut -c1-120
