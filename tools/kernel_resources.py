#!/usr/bin/env python3
"""Registers, LDS and occupancy the compiler reports for every kernel of one source file:
    python tools/kernel_resources.py phylomap_amd/csrc/phm_wtiles.hip [name-filter]
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950; the flags of the Makefile; EXTRA=-D... in the environment adds switches)."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("EXTRA", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln) or re.search(r"remark: .*?\bName: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"phm::\(anonymous namespace\)::|void ", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[\w/]+\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'LDS':>7s} {'waves/SIMD':>10s}")
for k, r in rows.items():
    if flt in k:
        print(f"{k[:58]:58s} {r.get('VGPRs', -1):5d} {r.get('AGPRs', -1):5d} {r.get('TotalSGPRs', -1):5d} {r.get('ScratchSize', -1):7d} "
              f"{r.get('LDS Size', -1):7d} {r.get('Occupancy', -1):10d}")
