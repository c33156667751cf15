#!/bin/bash
# Evidence pass v8: parity log, every bench workload with its CPU baseline, rocprofv3 kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run29
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
for w in var_p256 var_p384 msm_k256 var_k256 ecdsa_p256; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
prof() { name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OLDPWD/$OUT/prof_$name.log 2>&1)
}
prof fixed_k256 --steps 20 --warmup 3
prof msm_k256 --workload msm_k256 --steps 2 --warmup 1
prof var_p256 --workload var_p256 --steps 2 --warmup 1
prof var_p384 --workload var_p384 --steps 2 --warmup 1
prof ecdsa_p256 --workload ecdsa_p256 --steps 2 --warmup 1
python - <<'PY'
import csv, glob
for name in ("fixed_k256", "msm_k256", "var_p256", "var_p384", "ecdsa_p256"):
    fs = glob.glob("gpurun_out/run29/prof_%s/**/*kernel_stats.csv" % name, recursive=True)
    if not fs: continue
    print("--", name)
    for r in csv.DictReader(open(fs[0])):
        if "ecgpu" in r["Name"] and float(r["AverageNs"]) > 2e4:
            print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
