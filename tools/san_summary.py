#!/usr/bin/env python3
"""Summary of a sanitizer run of tools/gpu_run.sh (`san:asan` / `san:tsan`): the pytest verdict, the number of reports, and how many of
them have an access / a frame inside libecgpu (ours) against inside the uninstrumented HIP / HSA / RCCL runtimes.
    python tools/san_summary.py <out dir> <asan|tsan>"""
import glob
import os
import re
import sys

out, kind = sys.argv[1], sys.argv[2]
log = os.path.join(out, "san_%s.log" % kind)
tail = [ln for ln in open(log).read().splitlines() if ln.strip()][-3:] if os.path.exists(log) else ["(no pytest log)"]
print("pytest under %s: %s" % (kind, " | ".join(tail)))
reports = sorted(glob.glob(os.path.join(out, "san_%s_report*" % kind)))
total = ours = 0
kinds = {}
first_ours = None
for path in reports:
    txt = open(path, errors="replace").read()
    if kind == "tsan":
        blocks = [b for b in txt.split("==================\n") if "WARNING: ThreadSanitizer" in b]
    else:
        blocks = [b for b in re.split(r"(?==+\d+==ERROR|\S+:\d+:\d+: runtime error)", txt) if "ERROR: AddressSanitizer" in b or "runtime error" in b]
    for b in blocks:
        total += 1
        m = re.search(r"(?:ThreadSanitizer|AddressSanitizer): ([^(\n]*)|(runtime error: [^\n]*)", b)
        k = (m.group(1) or m.group(2)).strip() if m else "?"
        kinds[k] = kinds.get(k, 0) + 1
        accs = re.findall(r"\n  (?:Previous |Atomic |)?(?:[Rr]ead|[Ww]rite|atomic read|atomic write)[^\n]*\n((?:    #\d[^\n]*\n)+)", b) if kind == "tsan" else [b]
        mine = False
        for a in accs:
            fr = [f for f in a.strip().split("\n") if "libclang_rt" not in f and "libasan" not in f and "#" in f]
            if fr and "libecgpu" in fr[0]:
                mine = True
        if mine:
            ours += 1
            first_ours = first_ours or b[:3000]
print("%d report(s) in %d file(s): %s" % (total, len(reports), kinds))
print("with the reported access / first frame inside libecgpu_%s.so: %d" % (kind, ours))
if first_ours:
    print("---- first of them ----")
    print(first_ours)
