#!/bin/bash
# Evidence pass v9 for the fixed-base default (k256, W = 26): bench line, kernel stats, PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run33
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_fixed -o fixed -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/prof_fixed.log 2>&1)
pmc() { name=$1; shift; ctrs=""; while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --output-format csv -d $OLDPWD/$OUT/pmc_$name -o pmc -- "$@" > $OLDPWD/$OUT/pmc_$name.log 2>&1)
}
B="python $PWD/bench.py --no-cpu-baseline"
pmc valu_fixed SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B --steps 3 --warmup 1
pmc fetch_fixed FETCH_SIZE -- $B --steps 3 --warmup 1
pmc write_fixed WRITE_SIZE -- $B --steps 3 --warmup 1
python - <<'PY'
import csv, glob, collections
out = open("gpurun_out/run33/pmc_summary.txt", "w")
for d in sorted(glob.glob("gpurun_out/run33/pmc_*/")):
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
print(open("gpurun_out/run33/pmc_summary.txt").read())
for f in glob.glob("gpurun_out/run33/prof_fixed/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ecgpu" in r["Name"]:
            print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
echo done
