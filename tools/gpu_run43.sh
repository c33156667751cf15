#!/bin/bash
# Evidence pass v10 (XYZZ comb / bucket sums, two-level MSM sort): parity log, bench lines with CPU baselines, rocprofv3 kernel stats, PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run43
mkdir -p $OUT
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
for w in var_p256 var_p384 msm_k256 var_k256 ecdsa_p256 msm_p256; do
  echo "== bench $w"; timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_$w.json
done
prof() { name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$name -o $name -- python $OLDPWD/bench.py "$@" --no-cpu-baseline > $OLDPWD/$OUT/prof_$name.log 2>&1)
}
prof fixed --steps 20 --warmup 3
prof msm --workload msm_k256 --steps 2 --warmup 1
prof var_p256 --workload var_p256 --steps 2 --warmup 1
prof var_p384 --workload var_p384 --steps 2 --warmup 1
pmc() { name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d $OLDPWD/$OUT/pmc_$name -o pmc -- "$@" > $OLDPWD/$OUT/pmc_$name.log 2>&1)
}
B="python $PWD/bench.py --no-cpu-baseline"
pmc valu_fixed SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B --steps 3 --warmup 1
pmc fetch_fixed FETCH_SIZE -- $B --steps 3 --warmup 1
pmc write_fixed WRITE_SIZE -- $B --steps 3 --warmup 1
pmc valu_msm SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B --workload msm_k256 --steps 1 --warmup 1
pmc fetch_msm FETCH_SIZE -- $B --workload msm_k256 --steps 1 --warmup 1
pmc write_msm WRITE_SIZE -- $B --workload msm_k256 --steps 1 --warmup 1
python - <<'PY'
import csv, glob, collections, os
out = open("gpurun_out/run43/pmc_summary.txt", "w")
for d in sorted(glob.glob("gpurun_out/run43/pmc_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "ecgpu" not in k or "valu_probe" in k: continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    out.write("# %s\n" % d)
    for (k, c), v in sorted(acc.items()):
        out.write("%-62s %-22s n=%d avg=%.6g\n" % (k[:62], c, len(v), sum(v) / len(v)))
out.close()
print(open("gpurun_out/run43/pmc_summary.txt").read())
for name in ("fixed", "msm", "var_p256", "var_p384"):
    fs = glob.glob("gpurun_out/run43/prof_%s/**/*kernel_stats.csv" % name, recursive=True)
    if not fs: continue
    print("--", name)
    for r in csv.DictReader(open(fs[0])):
        if "ecgpu" in r["Name"]:
            print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT; echo done
