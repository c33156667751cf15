#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run12
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
for w in var_p256 var_p384 msm_k256; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_fixed -o fixed -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/prof_fixed.log 2>&1)
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/run12/prof_fixed/fixed_kernel_stats.csv")):
    if "ecgpu" in r["Name"]:
        print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
