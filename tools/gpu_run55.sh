#!/bin/bash
# Final check at the bp384 commit: smoke(), the default bench line, the N=1 variable-base and MSM lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_fixed_k256_v14.json
