#!/usr/bin/env python3
"""The fixed tail of the k256 MSM at one GPU's share (2^21 terms) and at the full size (2^24), for the two builds of the
latency-bound tail kernels (ECGPU_MSM_TAIL = 0: default flags, 1: scheduled for instruction-level parallelism).  Every
setting runs in a process of its own (the variant is read once) under rocprofv3 --kernel-trace so that each kernel's time is
listed.    python tools/gpu_msm_tail.py [sizes ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [21, 24]
for lg in sizes:
    for tail in ("0", "1"):
        env = dict(os.environ, ECGPU_MSM_TAIL=tail, TMPDIR="/tmp")
        out = "/tmp/msm_tail_%d_%s" % (lg, tail)
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--only", "msm_k256", "--n", str(1 << lg), "--steps", "10", "--warmup", "2", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("n=2^%d tail=%s FAILED\n%s" % (lg, tail, r.stderr[-2000:]))
            continue
        rec = json.loads(line[-1])
        print("n=2^%d ECGPU_MSM_TAIL=%s  %.3f ms/step  check=%s  stages=%s" % (
            lg, tail, rec["ms_per_step"], rec["check_vs_oracle"], {k: round(v, 3) for k, v in rec["stage_ms"].items()}), flush=True)
        st = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "stats", out], capture_output=True, text=True)
        for l in st.stdout.splitlines():
            if "k_msm" in l or "k_normalize<K256Params, 0>" in l:
                print("    " + l)
