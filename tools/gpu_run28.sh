#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run28
mkdir -p $OUT
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== bench var_p384"; timeout 900 python bench.py --workload var_p384 --steps 3 --warmup 1 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_var_p384.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_var_p384 -o var_p384 -- python $OLDPWD/bench.py --workload var_p384 --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_var_p384.log 2>&1)
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
echo done
