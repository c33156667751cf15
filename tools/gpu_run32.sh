#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python - <<'PY'
import importlib, time
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
for w in (24, 26):
    e.set_base_window(0, w)
    t0 = time.time(); e.mul_by_generator(0, bytes(31) + b"\x05"); t1 = time.time()
    print("k256 W=%d: table build + first call %.1f ms" % (w, (t1 - t0) * 1e3))
PY
for w in 24 26; do timeout 600 python bench.py --window $w --steps 20 --warmup 3 --check --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fixed W=$w', '%.4g'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'kernel %.3f'%d['roofline']['kernel_ms'], d.get('check_vs_oracle'))"; done
