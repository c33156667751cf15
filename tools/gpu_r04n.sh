#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  echo "== events on";  python tools/gpu_fixed_overlap.py 2>&1 | grep "1 in flight\|rror" | head -8
  echo "== events off"; FIXED_TIMING=0 python tools/gpu_fixed_overlap.py 2>&1 | grep "in flight\|rror" | head -12
done
