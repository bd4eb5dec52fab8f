#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 3 1; do
  python3 tools/scratch/item_major.py $B 20
  out=gpurun_out/im_$B; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/scratch/item_major.py $B 10 > $out/log.txt 2>&1
  st=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$st" $B <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2]); n = 13 * (1 if B == 3 else 3)
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"  sum of kernel durations per 3 items: {tot / 13 / 1e3:.0f} us")
for r in rows[:9]:
    print(f"   {int(r['TotalDurationNs'])/13/1e3:8.1f} us/3 items  max {float(r['MaxNs'])/1e3:7.1f}  {r['Name'][:90]}")
PY
  find $out -name '*kernel_trace.csv' -delete
done
