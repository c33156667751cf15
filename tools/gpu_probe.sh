#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/probe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/probe/prof -- python $GRAFT_REPO_ROOT/bench.py --workload msm_k256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/probe/prof -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msm' in r['Name']:
        print(r['Name'].split('(')[0][-45:], r['Calls'], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
rm -rf gpurun_out/probe/prof
