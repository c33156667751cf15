#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/small_msm
mkdir -p $OUT
for n in 4096 65536; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$n -o p -- python $OLDPWD/bench.py --workload msm_k256 --n $n --steps 20 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof_$n.log 2>&1)
tail -1 $OUT/prof_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n', d['config']['units_per_gpu'], 'ms/step %.3f'%d['ms_per_step'])"
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/small_msm/prof_$n/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msm" in r["Name"] or "normalize" in r["Name"]:
            print("  %-52s calls=%-3s avg_us=%.1f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $OUT -name "*.db" -delete
