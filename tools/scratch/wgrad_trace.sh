#!/bin/bash
# per-launch durations of wgrad_f16_kernel in one pre-training step (single stream), in launch order, with grid sizes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/wg_trace
rm -rf "$out"; mkdir -p "$out"
US_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 bench_pretrain.py --iters 2 --warmup 1 "$@" > "$out/bench.log" 2>&1
tr=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$tr" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_f16_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 69
last = rows[-n:]
tot = 0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{i:3d} {d:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')}  wg {r.get('Workgroup_Size_X','')}")
print("total", tot)
PY
find "$out" -name '*kernel_trace.csv' -delete
