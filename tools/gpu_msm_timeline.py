#!/usr/bin/env python3
"""Start / end of every kernel of ONE k256 MSM step (of a short bench run) relative to the step's first kernel, from
rocprofv3 --kernel-trace: which kernel waits for which, and the gaps between them.
    python tools/gpu_msm_timeline.py [log2 n]"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out = "/tmp/msm_timeline_%d" % lg
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable, os.path.join(ROOT, "bench.py"),
       "--only", "msm_k256", "--n", str(1 << lg), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
rows = []
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ecgpu::", ""),
                     r.get("Queue_Id", "?"), r.get("Grid_Size", "?")))
rows.sort()
# the last step: from the last k_msm_prepare on
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_msm_prepare")]
if not starts:
    sys.exit("no MSM kernels in the trace")
# the check of the bench runs more MSMs after the timed steps; take the last timed one = the (warmup + steps)-th prepare
idx = starts[min(len(starts) - 1, 3)]
end = starts[starts.index(idx) + 1] if starts.index(idx) + 1 < len(starts) else len(rows)
t0 = rows[idx][0]
print("n = 2^%d: one step, times in ms from the start of k_msm_prepare" % lg)
prev_end, busy, gaps = None, 0, 0
for s, e, name, q, grid in rows[idx:end]:
    gap = 0 if prev_end is None else s - prev_end
    print("  %9.3f .. %9.3f  (%7.3f)  gap before %6.1f us  queue %-3s grid %-9s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, gap / 1e3, q, grid, name[:60]))
    if name.startswith("k_msm_combine"):
        busy += e - s
        gaps += max(gap, 0)
        print("  step: %.3f ms from the first kernel's start to the last one's end = %.3f ms in kernels + %.3f ms between them (%d launches)" % (
            (e - t0) / 1e6, busy / 1e6, gaps / 1e6, rows[idx:end].index((s, e, name, q, grid)) + 1))
        break
    busy += e - s
    gaps += max(gap, 0)
    prev_end = max(e, prev_end or 0)
