#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one .hip file (the compiler's view:
-Rpass-analysis=kernel-resource-usage).      python tools/kernel_resources.py cnc_amd/csrc/field_fused.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "-munsafe-fp-atomics", "-c", "-I", "include", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print("%-80s vgpr %4d agpr %4d spill %4d sgpr %4d scratch %5d occ %d" % (
        name[:80], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("TotalSGPRs", -1),
        r.get("ScratchSize", -1), r.get("Occupancy", -1)))
