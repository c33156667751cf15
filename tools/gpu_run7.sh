#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run7
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
for w in var_p256 var_p384; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_var -o var -- python $OLDPWD/bench.py --workload var_p256 --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_var.log 2>&1)
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/run7/prof_var/var_kernel_stats.csv")):
    if "ecgpu" in r["Name"]:
        print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
