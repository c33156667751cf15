#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/run42
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/run42/pytest_gpu.txt
python - <<'PY' | tee gpurun_out/run42/host_pointer_rates.txt
import importlib, time, numpy as np
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
rng = np.random.default_rng(1)
for lg in (20, 22):
    n = 1 << lg
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8); k[:, 0] &= 0x7f
    out, inf = np.zeros(n * 64, np.uint8), np.zeros(n, np.uint8)
    ps, po, pi = e.host_alloc(n * 32), e.host_alloc(n * 64), e.host_alloc(n)
    ps[:] = k.reshape(-1)
    for name, a in (("pageable", (k.reshape(-1), out, inf)), ("page-locked", (ps, po, pi))):
        e.mul_by_generator(0, a[0], out=a[1], inf=a[2])
        t = time.perf_counter(); reps = 5
        for _ in range(reps): e.mul_by_generator(0, a[0], out=a[1], inf=a[2])
        dt = (time.perf_counter() - t) / reps
        print("host-pointer fixed-base k256 2^%d, %s buffers: %.2f ms -> %.3e /s (device part %.2f ms)" % (lg, name, dt * 1e3, n / dt, e.last_timing("total")))
    assert bytes(out) == bytes(po)
    for a in (ps, po, pi): e.host_free(a)
PY
