#!/bin/bash
# Third GPU pass: the unsaturated-limb field arithmetic. Parity, bench lines, kernel stats, PMC.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run3
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
for w in 12 14; do echo "== bench fixed window $w"; timeout 600 python bench.py --steps 5 --warmup 1 --window $w --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_w$w.json; done
echo "== bench msm 2^24"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
echo "== bench msm 2^24 c=15"; timeout 900 python bench.py --workload msm_k256 --steps 2 --warmup 1 --window 15 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256_c15.json
echo "== bench msm 2^20"; timeout 600 python bench.py --workload msm_k256 --n 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_2p20.json
echo "== bench var p256"; timeout 900 python bench.py --workload var_p256 --steps 3 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_var_p256.json
echo "== bench var p384"; timeout 900 python bench.py --workload var_p384 --steps 2 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_var_p384.json
prof() { # name, bench args
  name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OLDPWD/$OUT/prof_$name.log 2>&1)
  for f in $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); do echo "-- $name"; cut -c1-150 $f | head -9; done
}
prof fixed --steps 5 --warmup 1
prof msm --workload msm_k256 --steps 2 --warmup 1
prof var_p256 --workload var_p256 --steps 2 --warmup 1
pmc() { # name, counters..., -- bench args
  name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d $OLDPWD/$OUT/pmc_$name -o pmc -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OLDPWD/$OUT/pmc_$name.log 2>&1)
}
pmc valu_fixed SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --steps 3 --warmup 1
pmc fetch_fixed FETCH_SIZE -- --steps 3 --warmup 1
pmc write_fixed WRITE_SIZE -- --steps 3 --warmup 1
pmc valu_msm SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --workload msm_k256 --steps 1 --warmup 1
pmc fetch_msm FETCH_SIZE -- --workload msm_k256 --steps 1 --warmup 1
pmc write_msm WRITE_SIZE -- --workload msm_k256 --steps 1 --warmup 1
python - <<'PY'
import csv, glob, collections, os
out = open(os.path.join("gpurun_out/run3", "pmc_summary.txt"), "w")
for d in sorted(glob.glob("gpurun_out/run3/pmc_*/")):
    fs = glob.glob(d + "*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "ecgpu" not in k or "probe" in k: continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    out.write("# %s\n" % d)
    for (k, c), v in sorted(acc.items()):
        out.write("%-62s %-22s n=%d avg=%.6g\n" % (k[:62], c, len(v), sum(v) / len(v)))
out.close()
print(open("gpurun_out/run3/pmc_summary.txt").read())
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT; echo done
