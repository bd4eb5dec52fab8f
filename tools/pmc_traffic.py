#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of bench.py into the
per-launch HBM traffic of the dominant kernel, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane)
coalesced read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores (the conv epilogue stores
4 B per lane, 128-B segments: uncalibrated, reported as measured).
usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import csv
import json
import sys


def per_kernel(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if "conv_igemm" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"kernel": "conv_igemm_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_kib_per_launch_raw": fetch / max(nf, 1), "write_kib_per_launch": write / max(nw, 1),
       "hbm_bytes_per_launch": (2.0 * fetch / max(nf, 1) + write / max(nw, 1)) * 1024.0,
       "note": "FETCH_SIZE doubled (gfx950 128-B request counted as 64 B); includes Infinity-Cache hits (memory-side counters)"}
out = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
