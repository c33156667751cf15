#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run9
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== bench default"; timeout 900 python bench.py --steps 10 --warmup 2 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256.json
echo "== bench msm"; timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256.json
for ch in 512 700 1024 2048; do echo "== bench msm chunk $ch"; ECGPU_MSM_CHUNK=$ch timeout 900 python bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_k256_chunk$ch.json; done
echo "== bench msm 2^16 check"; timeout 600 python bench.py --workload msm_k256 --n 65536 --steps 2 --warmup 1 --no-cpu-baseline --check 2>&1 | tail -1 | tee $OUT/bench_msm_2p16_check.json
echo "== bench msm 2^20"; timeout 600 python bench.py --workload msm_k256 --n 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_msm_2p20.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_msm -o msm -- python $OLDPWD/bench.py --workload msm_k256 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_msm.log 2>&1)
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/run9/prof_msm/msm_kernel_stats.csv")):
    if "ecgpu" in r["Name"]:
        print("  %-52s calls=%-3s avg_ms=%.3f" % (r["Name"].split("(")[0][:52], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
