#!/usr/bin/env python3
"""Compressed points INTO the path, measured: the 2^24-term k256 MSM and the 2^20-pair p256 / k256 batch
multiplication handed over in page-locked HOST memory as x || y records (ecgpu_msm / ecgpu_batch_mul) and as x + tag records
(ecgpu_msm_compressed / ecgpu_batch_mul_compressed), plus the device-resident rates of the decoding alone.  The compressed
form ships 65 instead of 96 bytes per term but pays one square root per point on the device.
    python tools/gpu_compressed_e2e.py [log2 terms of the MSM]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
from gpu_common import rand_scalars  # noqa: E402

eng = ec.Engine(0)


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def setup(curve, n):
    cid = ec.CURVE_IDS[curve]
    L = ec.FIELD_BYTES[cid]
    d_s = eng.to_device(rand_scalars(cid, n, 0x5EC1))
    d_k = eng.to_device(rand_scalars(cid, n, 0x5EC2))
    d_p, d_f = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n)
    eng.mul_by_generator_dev(cid, d_s, n, d_p, d_f)
    P = eng.to_host(d_p).reshape(n, 2 * L)
    h_k, h_p, h_x, h_t = eng.host_alloc(n * L), eng.host_alloc(n * 2 * L), eng.host_alloc(n * L), eng.host_alloc(n)
    h_k[:] = eng.to_host(d_k)
    h_p[:] = P.reshape(-1)
    h_x[:] = P[:, :L].reshape(-1)
    h_t[:] = 2 + (P[:, 2 * L - 1] & 1)
    d_x, d_t = eng.to_device(h_x), eng.to_device(h_t)
    return cid, L, (d_k, d_p, d_x, d_t), (h_k, h_p, h_x, h_t)


lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
cid, L, (d_k, d_p, d_x, d_t), (h_k, h_p, h_x, h_t) = setup("k256", n)
d_o, d_i = eng.dev_alloc(2 * L + 64), eng.dev_alloc(16)
t_dev, _ = best(lambda: eng.lincomb_dev(cid, d_k, d_p, None, n, d_o, d_i))
ref = bytes(eng.to_host(d_o, 2 * L))
t_devc, _ = best(lambda: eng.lincomb_compressed_dev(cid, d_k, d_x, d_t, n, d_o, d_i))
same = bytes(eng.to_host(d_o, 2 * L)) == ref
t_h, r1 = best(lambda: eng.lincomb(cid, h_k, h_p))
t_hc, r2 = best(lambda: eng.lincomb_compressed(cid, h_k, h_x, h_t))
same = same and bytes(r1[0]) == ref and bytes(r2[0]) == ref
d_xy, d_ok = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n)
lib = ec.load_library()
import ctypes  # noqa: E402
dp = ec._dp
t_dec, _ = best(lambda: lib.ecgpu_batch_decompress_dev(eng._ctx, cid, dp(d_x), dp(d_t), ctypes.c_size_t(n), dp(d_xy), dp(d_ok)))
print("k256 MSM, 2^%d terms: device-resident x||y %.2f ms, x+tag %.2f ms (decoding alone: %.2f ms = %.1f ns per point); "
      "from pinned host memory x||y (%.2f GB) %.2f ms, x+tag (%.2f GB) %.2f ms; results equal: %s" % (
          lg, t_dev, t_devc, t_dec, t_dec * 1e6 / n, n * 3 * L / 1e9, t_h, n * (2 * L + 1) / 1e9, t_hc, same), flush=True)
for b in (d_k, d_p, d_x, d_t, d_o, d_i, d_xy, d_ok):
    b.free()
for h in (h_k, h_p, h_x, h_t):
    eng.host_free(h)

for curve in ("p256", "k256"):
    n = 1 << 20
    cid, L, (d_k, d_p, d_x, d_t), (h_k, h_p, h_x, h_t) = setup(curve, n)
    d_q, d_f = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n)
    t_dev, _ = best(lambda: eng.mul_dev(cid, d_k, d_p, None, n, d_q, d_f))
    ref = bytes(eng.to_host(d_q))
    t_devc, _ = best(lambda: eng.mul_compressed_dev(cid, d_k, d_x, d_t, n, d_q, d_f))
    same = bytes(eng.to_host(d_q)) == ref
    t_h, r1 = best(lambda: eng.mul(cid, h_k, h_p))
    t_hc, r2 = best(lambda: eng.mul_compressed(cid, h_k, h_x, h_t))
    same = same and bytes(r1[0]) == ref and bytes(r2[0]) == ref
    print("%s batch mul, 2^20 pairs: device-resident x||y %.2f ms, x+tag %.2f ms; from pinned host memory x||y %.2f ms, x+tag %.2f ms; "
          "results equal: %s" % (curve, t_dev, t_devc, t_h, t_hc, same), flush=True)
    for b in (d_k, d_p, d_x, d_t, d_q, d_f):
        b.free()
    for h in (h_k, h_p, h_x, h_t):
        eng.host_free(h)
eng.close()
