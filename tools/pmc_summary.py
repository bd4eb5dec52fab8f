#!/usr/bin/env python3
"""Join the rocprofv3 --pmc passes of tools/pmc_collect.sh: per kernel instantiation, and for conv_igemm_kernel per U-Net launch
(dispatch order of one evaluation, same schedule as tools/analyze_trace.py), sums of every counter plus the derived figures.

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a 128-B
request of a wide (16 B/lane) read as 64 B, so it is doubled; both are memory-side (Infinity-Cache hits included).  SQ_WAVE_CYCLES,
SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles.
usage: tools/pmc_summary.py <dir with sq.csv fetch.csv write.csv [mfma.csv grbm.csv]> <out.json>"""
import collections
import csv
import json
import os
import subprocess
import sys

d, out = sys.argv[1], sys.argv[2]


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("us::", "")
    return n.split("(")[0]


per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
conv_rows = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch order index -> counter -> value
for fn in ("sq", "mfma", "fetch", "write", "grbm"):
    path = os.path.join(d, fn + ".csv")
    if not os.path.exists(path):
        continue
    seen = set()
    conv_i = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        per_kernel[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r["Dispatch_Id"]
        if fn == "sq" and (k, did) not in seen:
            seen.add((k, did))
            calls[k] += 1
        if "conv_igemm_kernel" in k:
            if did not in conv_i:
                conv_i[did] = len(conv_i)
            conv_rows[conv_i[did]][r["Counter_Name"]] += float(r["Counter_Value"])
            conv_rows[conv_i[did]]["_kernel"] = k

res = {"kernels": {}, "notes": __doc__.split("usage")[0].strip()}
try:
    res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    res["commit"] = os.environ.get("US_COMMIT")


try:      # fingerprint of the inference kernels' sources: bench.py refuses this summary's traffic figures on any other tree
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as _bench
    from unitspeech_amd._build import source_fingerprint
    res["source_sha256"] = source_fingerprint(_bench.TRAFFIC_SOURCES)
except Exception as e:      # noqa: BLE001
    res["source_sha256"] = None
    print("no source fingerprint:", e)


def derive(c, n):
    o = {"dispatches": n}
    o.update({k: v for k, v in c.items() if not k.startswith("_")})
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        o["wave_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = c.get("SQ_WAIT_ANY", 0) / wc
        o["issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
        o["issuing_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = c.get("SQ_ACTIVE_INST_ANY", 0) / wc
    if c.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is per SE-level SQ; MFMA busy is summed over SIMDs: report the raw ratio and leave scaling to the reader
        o["mfma_busy_cycles_per_sq_busy_cycle"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        o["hbm_side_bytes_per_dispatch (2*FETCH_SIZE + WRITE_SIZE, KiB -> B)"] = (2.0 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / max(n, 1)
    return o


for k, c in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    res["kernels"][k] = derive(c, calls.get(k, 0))

# per U-Net launch of one evaluation: the trace holds E evaluations x n launches in schedule order
n = len(conv_rows)
sched = None
try:
    import importlib.util
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "analyze_trace.py")).read()
    # reuse the launch schedule builder of analyze_trace.py (everything before it reads the trace)
    head = src.split("rows = [r for r in csv.DictReader")[0].replace("a = ap.parse_args()", "a = ap.parse_args(['x'])")
    ns = {}
    exec(compile(head, "analyze_trace_head", "exec"), ns)
    sched = [s[0] for s in ns["seq"]]
except Exception as e:      # schedule unavailable: keep the per-kernel part
    res["per_launch_error"] = str(e)
if sched and n % len(sched) == 0 and n:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    kern = {}
    for i in range(n):
        j = i % len(sched)
        cnt[j] += 1
        for ck, v in conv_rows[i].items():
            if ck == "_kernel":
                kern[j] = v
            else:
                per[j][ck] += v
    res["conv_launches"] = []
    tot_bytes = 0.0
    for j, name in enumerate(sched):
        o = derive(per[j], cnt[j])
        o["launch"] = name
        o["kernel"] = kern.get(j)
        tot_bytes += o.get("hbm_side_bytes_per_dispatch (2*FETCH_SIZE + WRITE_SIZE, KiB -> B)", 0.0)
        res["conv_launches"].append(o)
    res["hbm_bytes_per_launch"] = tot_bytes / len(sched)
    res["conv_hbm_side_bytes_per_evaluation"] = tot_bytes
# every kernel of the score network, per evaluation: the trace holds as many evaluations as final_conv_kernel dispatches; weight packs,
# table copies and torch's own kernels (set-up, once per handle) are left out
n_eval = max(sum(v for k, v in calls.items() if k.startswith("final_conv_kernel")), 1)      # (a template since round 3: final_conv_kernel<true>)
setup = ("pack_", "copy_table", "copy_rows", "fill_", "at::", "__amd_rocclr", "l2_normalize", "mul_mask", "finish_mel", "Cijk")
all_bytes = 0.0
for k, o in res["kernels"].items():
    if any(t in k for t in setup):
        continue
    all_bytes += o.get("hbm_side_bytes_per_dispatch (2*FETCH_SIZE + WRITE_SIZE, KiB -> B)", 0.0) * o.get("dispatches", 0)
res["evaluations_in_trace"] = n_eval
res["all_kernels_hbm_side_bytes_per_evaluation"] = all_bytes / n_eval
json.dump(res, open(out, "w"), indent=1)
print(f"all kernels of one evaluation: {all_bytes / n_eval / 1e9:.2f} GB HBM-side ({n_eval} evaluations in the trace)")
print(f"{len(res['kernels'])} kernels; conv launches per evaluation: {len(sched) if sched else '?'}; "
      f"conv HBM-side bytes per evaluation: {res.get('conv_hbm_side_bytes_per_evaluation', 0) / 1e9:.2f} GB")
for k, o in list(res["kernels"].items())[:8]:
    print(f"  {k[:70]:70s} parked {o.get('wave_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)', 0):.2f} "
          f"stall {o.get('issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)', 0):.2f} "
          f"ldsconf {o.get('lds_bank_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)', 0):.3f} "
          f"bytes/disp {o.get('hbm_side_bytes_per_dispatch (2*FETCH_SIZE + WRITE_SIZE, KiB -> B)', 0) / 1e6:.0f} MB")
