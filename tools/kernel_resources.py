#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage
(runs here, no GPU needed): python tools/kernel_resources.py unitspeech_amd/csrc/conv_igemm.hip [extra hipcc flags]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>6} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
    for c in rows:
        print(f"{c.get('VGPRs', 0):5d} {c.get('AGPRs', 0):5d} {c.get('VGPRs Spill', 0):6d} {c.get('ScratchSize', 0):8d} "
              f"{c.get('Occupancy', 0):4d} {c.get('LDS Size', 0):7d}  {c['name'][:150]}")


if __name__ == "__main__":
    main()
