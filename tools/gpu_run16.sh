#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/run16
mkdir -p $OUT
python - <<'PY'
import importlib, time, numpy as np
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
for w in (20, 22, 24):
    for cid, L in ((0, 32), (1, 32)):
        e.set_base_window(cid, w)
        t0 = time.time(); e.mul_by_generator(cid, bytes(L - 1) + b"\x05"); t1 = time.time()
        print("W=%d curve %d table build + first call %.1f ms" % (w, cid, (t1 - t0) * 1e3), flush=True)
PY
for w in 20 22 23 24; do echo "== bench fixed W=$w"; timeout 600 python bench.py --window $w --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_fixed_k256_w$w.json; done
echo done
