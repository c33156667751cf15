#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run5
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
echo "== bench var p384"; timeout 900 python bench.py --workload var_p384 --steps 2 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_var_p384.json
echo "== bench var p256"; timeout 900 python bench.py --workload var_p256 --steps 3 --warmup 1 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_var_p256.json
for t in 18 20 21; do echo "== bench msm tile 2^$t"; ECGPU_MSM_TILE_LOG2=$t timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256_tile$t.json; done
prof() { name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OLDPWD/$OUT/prof_$name.log 2>&1)
  python - $OUT $name <<'PY'
import csv, sys
out, name = sys.argv[1], sys.argv[2]
print("--", name)
for r in csv.DictReader(open("%s/prof_%s/%s_kernel_stats.csv" % (out, name, name))):
    if "ecgpu" in r["Name"]:
        print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
}
prof fixed --steps 5 --warmup 1
prof var_p384 --workload var_p384 --steps 2 --warmup 1
ECGPU_MSM_TILE_LOG2=20 prof msm --workload msm_k256 --steps 2 --warmup 1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
