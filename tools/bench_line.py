#!/usr/bin/env python3
"""One-line digest of bench.py JSON lines: tools/bench_line.py <log> [<log> ...]"""
import json
import sys

for path in sys.argv[1:]:
    lines = [x for x in open(path) if x.startswith("{")]
    if not lines:
        print(path, "NO JSON:", open(path).read()[-800:])
        continue
    d = json.loads(lines[-1])
    r = d.get("roofline", {})
    print(f"{path}: {d['value']:.1f} {d.get('unit', '')}  ms/step {d.get('ms_per_step', 0):.2f} (median {d.get('ms_per_step_median', 0):.2f})  "
          f"frac {r.get('frac', 0):.4f}  eval {r.get('sampled_eval_ms', 0):.3f} ms  conv share {r.get('conv_share_of_eval_time', 0):.3f}")
