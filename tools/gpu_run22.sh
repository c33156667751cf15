#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run22
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== bench ecdsa"; timeout 900 python bench.py --workload ecdsa_p256 --steps 3 --warmup 1 --check 2>&1 | tail -3 | tee $OUT/bench_ecdsa_p256.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_ecdsa -o ecdsa -- python $OLDPWD/bench.py --workload ecdsa_p256 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_ecdsa.log 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/run22/prof_ecdsa/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ecgpu" in r["Name"]:
            print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
