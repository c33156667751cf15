#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run11
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -14 | tee $OUT/pytest_gpu.txt
echo "== table build time"; python - <<'PY'
import importlib, time, numpy as np
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
for cid, L in ((0, 32), (1, 32), (2, 48)):
    t0 = time.time(); e.mul_by_generator(cid, bytes(L - 1) + b"\x05"); t1 = time.time()
    e.mul_by_generator(cid, bytes(L - 1) + b"\x05"); t2 = time.time()
    print("curve %d first call %.1f ms, second %.2f ms" % (cid, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
PY
echo "== bench default"; timeout 900 python bench.py --steps 20 --warmup 3 --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_fixed -o fixed -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/prof_fixed.log 2>&1)
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/run11/prof_fixed/fixed_kernel_stats.csv")):
    if "ecgpu" in r["Name"]:
        print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
