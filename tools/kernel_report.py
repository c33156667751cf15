#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: name, VGPR, AGPR, scratch, occupancy, LDS."""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
rows = []
cur = {}
for line in txt.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|Dynamic Stack): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": v}
    else:
        cur[k if k=="VGPRs Spill" else k.split(" ")[0]] = v
if cur: rows.append(cur)
def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n
print("%-70s %5s %5s %8s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "vspill", "occ", "LDS"))
for r in rows:
    print("%-70s %5s %5s %8s %6s %4s %7s" % (dem(r["name"])[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("Occupancy"), r.get("LDS")))
