#!/usr/bin/env python3
"""The k256 MSM at one GPU's share (2^21 terms) and at the full size (2^24): step time without a profiler, then the
per-kernel times under rocprofv3 --kernel-trace.  Every size runs in a process of its own; extra environment settings
(tuning knobs such as ECGPU_MSM_CHUNK, ECGPU_MSM_SEG: with ECGPU_TOOL_LIB=.../lib/libecgpu_knobs.so, the build that reads them) are inherited.    python tools/gpu_msm_stages.py [log2 sizes ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [21, 24]
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--only", "msm_k256", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]


def line(out):
    l = [x for x in out.splitlines() if x.startswith("{")]
    return json.loads(l[-1]) if l else None


for lg in sizes:
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(BENCH + ["--n", str(1 << lg)], env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    rec = line(r.stdout)
    if not rec:
        print("n=2^%d FAILED\n%s" % (lg, r.stderr[-2000:]))
        continue
    print("n=2^%d  %.3f ms/step  check=%s  stages=%s" % (
        lg, rec["ms_per_step"], rec["check_vs_oracle"], {k: round(v, 3) for k, v in rec["stage_ms"].items()}), flush=True)
    out = "/tmp/msm_stages_%d" % lg
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--"] + BENCH + ["--n", str(1 << lg)]
    r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    rec = line(r.stdout)
    if rec:
        print("    under the profiler: %.3f ms/step" % rec["ms_per_step"])
    st = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "stats", out], capture_output=True, text=True)
    for l in st.stdout.splitlines():
        if "k_msm" in l or "k_normalize<K256Params, 0>" in l:
            print("    " + l)
