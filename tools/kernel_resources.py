#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]  # further arguments go to hipcc (e.g. -DRSEM_Q32_DEPTHS=4,4,3,3)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-w",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in out.split("\n"):
    m = re.search(r"remark:\s+(Function Name|Name): (\S+)", line)
    if m:
        cur = m.group(2)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if "rocprim" in k or "hipcub" in k:
        continue
    name = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    print("%-40s VGPR %3d AGPR %3d SGPR %3d scratch %4d LDS %6d occ %d" % (
        name[:40], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize [bytes/lane]", -1),
        v.get("LDS Size [bytes/block]", -1), v.get("Occupancy [waves/SIMD]", -1)))
