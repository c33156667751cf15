#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python - <<'PY'
import importlib, time
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
for cid, L in ((0, 32), (1, 32), (2, 48)):
    t0 = time.time(); e.mul_by_generator(cid, bytes(L - 1) + b"\x05"); t1 = time.time()
    print("curve %d: table build + first call %.1f ms" % (cid, (t1 - t0) * 1e3))
PY
timeout 600 python bench.py --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed', '%.4g'%d['value'], d.get('check_vs_oracle'))"
