#!/usr/bin/env python3
"""Rate of batch bign verification (ecgpu_bign_verify_batch_dev / _msg_batch_dev) on 2^20 device-resident signatures.
Inputs: 2^14 distinct valid signatures — keys and nonce points by the engine's own generator multiplication, the challenge hash by
the oracle's belt-hash (input preparation only), S1 in Python integers — tiled to 2^20; every verdict is checked (all valid, and
all invalid after one byte of every hash is changed).    python tools/gpu_bign_rate.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
import oracle_lib  # noqa: E402
import pyec  # noqa: E402
from gpu_common import rand_scalars  # noqa: E402

c = pyec.CURVES["bign256"]
cid = c.cid
e = ec.Engine(0)
m, n, msg_len = 1 << 14, 1 << 20, 64
d = rand_scalars(cid, m, 0xB1A0)
k = rand_scalars(cid, m, 0xB1A1)
q, _ = e.mul_by_generator(cid, d)
r, _ = e.mul_by_generator(cid, k)
rng = np.random.default_rng(0xB1A2)
msgs = rng.integers(0, 256, m * msg_len, dtype=np.uint8).tobytes()
q, r, d, k = bytes(q), bytes(r), bytes(d), bytes(k)
hs, sigs = bytearray(), bytearray()
for i in range(m):
    h = oracle_lib.belt_hash(msgs[i * msg_len:(i + 1) * msg_len])
    s0 = oracle_lib.belt_hash(pyec.BELT_OID + r[64 * i:64 * i + 32] + h)[:16]
    di, ki = int.from_bytes(d[32 * i:32 * i + 32], "little"), int.from_bytes(k[32 * i:32 * i + 32], "little")
    s1 = (ki - int.from_bytes(h, "little") - (int.from_bytes(s0, "little") + 2 ** 128) * di) % c.n
    hs += h
    sigs += s0 + s1.to_bytes(32, "little")
reps = n // m
d_h, d_s, d_q, d_m = (e.to_device(bytes(x) * reps) for x in (hs, sigs, q, msgs))
d_ok = e.dev_alloc(n)
for name, call in (("prehash", lambda: e.bign_verify_dev(d_h, d_s, d_q, n, d_ok)),
                   ("message (%d bytes, belt-hash on the device)" % msg_len, lambda: e.bign_verify_msg_dev(d_q, d_m, msg_len, d_s, n, d_ok))):
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        call()
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None or ms < best else best
    ok = e.to_host(d_ok, n)
    st = {s: round(e.last_timing(s) or 0, 3) for s in ("recode", "main", "normalize", "total")}
    print("bign verify, %-46s 2^20 signatures: %.2f ms -> %.3e /s   all valid: %s   stages %s" % (name, best, n / best * 1e3, bool(ok.all()), st), flush=True)
bad = np.frombuffer(bytes(hs) * reps, np.uint8).copy()
bad[::32] ^= 1
e.bign_verify_dev(e.to_device(bad), d_s, d_q, n, d_ok)
print("every hash changed in one bit: verdicts all 0: %s" % (not e.to_host(d_ok, n).any()))
e.close()
