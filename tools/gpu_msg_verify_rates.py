"""Device-resident rates of the message-level verification entry points (digest on the device) beside the prehash ones:
2^20 signatures per call, 2^12 distinct ones tiled.  python tools/gpu_msg_verify_rates.py"""
import hashlib, importlib, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyec
from gpu_common import rand_scalars
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
n, m = 1 << 20, 1 << 9
H = {"k256": "sha256", "p256": "sha256", "p384": "sha384"}


def timed(fn, reps=5):
    fn(); e.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    e.synchronize()
    return (time.perf_counter() - t0) / reps


for curve in ("k256", "p256", "p384", "sm2"):
    c = pyec.CURVES[curve]
    L = c.L
    for msg_len in (32, 256):
        rng = random.Random(5 + msg_len)
        ds, ks = rand_scalars(c.cid, m, 0xA0 + c.cid), rand_scalars(c.cid, m, 0xA1 + c.cid)
        Q, _ = e.mul_by_generator(c.cid, ds)
        R, _ = e.mul_by_generator(c.cid, ks)
        msgs = np.frombuffer(bytes(rng.randrange(256) for _ in range(m * msg_len)), np.uint8)
        zs, sg = bytearray(), bytearray()
        ident = b"1234567812345678"
        for i in range(m):
            d = int.from_bytes(bytes(ds[i * L:(i + 1) * L]), "big"); k = int.from_bytes(bytes(ks[i * L:(i + 1) * L]), "big") or 1
            mb = bytes(msgs[i * msg_len:(i + 1) * msg_len])
            if curve == "sm2":
                Qi = (int.from_bytes(bytes(Q[2 * L * i: 2 * L * i + L]), "big"), int.from_bytes(bytes(Q[2 * L * i + L: 2 * L * (i + 1)]), "big"))
                ev = int.from_bytes(hashlib.new("sm3", pyec.sm2_za(c, ident, Qi) + mb).digest(), "big")
                sig = None
                kk = k
                while sig is None:
                    sig = pyec.sm2dsa_sign(c, d, ev, kk); kk += 1
                r, s_ = sig
                zs += ev.to_bytes(L, "big")
            else:
                dg = hashlib.new(H[curve], mb).digest()
                z = int.from_bytes(dg[:L], "big")
                x = int.from_bytes(bytes(R[2 * L * i: 2 * L * i + L]), "big")
                r = x % c.n
                s_ = pow(k, -1, c.n) * (z + r * d) % c.n
                if curve == "k256" and s_ > c.n // 2:
                    s_ = c.n - s_
                zs += dg[:L]
            sg += r.to_bytes(L, "big") + s_.to_bytes(L, "big")
        tile = lambda b, unit: np.ascontiguousarray(np.tile(np.frombuffer(bytes(b), np.uint8).reshape(m, unit), (n // m, 1))).reshape(-1)
        d_q, d_m, d_s = e.to_device(tile(Q, 2 * L)), e.to_device(tile(msgs, msg_len)), e.to_device(tile(sg, 2 * L))
        d_ok = e.dev_alloc(n + 16)
        sgb = np.frombuffer(bytes(sg), np.uint8).reshape(m, 2 * L)
        d_z, d_r, d_ss = e.to_device(tile(zs, L)), e.to_device(tile(sgb[:, :L].copy(), L)), e.to_device(tile(sgb[:, L:].copy(), L))
        e.set_async(True)
        if curve == "sm2":
            d_id = e.to_device(np.frombuffer(ident, np.uint8))
            t_msg = timed(lambda: e.sm2dsa_verify_msg_dev(d_id, len(ident), d_q, d_m, msg_len, d_s, n, d_ok))
            ok_msg = int(e.to_host(d_ok, n).sum())
            t_pre = timed(lambda: e._chk(e._lib.ecgpu_sm2dsa_verify_batch_dev(e._ctx, ec._dp(d_z), ec._dp(d_r), ec._dp(d_ss), ec._dp(d_q), ec.ctypes.c_size_t(n), ec._dp(d_ok))))
        else:
            t_msg = timed(lambda: e.ecdsa_verify_msg_dev(c.cid, d_q, d_m, msg_len, d_s, n, curve == "k256", d_ok))
            ok_msg = int(e.to_host(d_ok, n).sum())
            t_pre = timed(lambda: e.ecdsa_verify_dev(c.cid, d_z, d_r, d_ss, d_q, n, curve == "k256", d_ok))
        ok_pre = int(e.to_host(d_ok, n).sum())
        e.set_async(False)
        print("%-5s %3d-byte messages: from messages %.2f ms (%.2e /s), from prehashes %.2f ms (%.2e /s); accepted %d / %d of %d"
              % (curve, msg_len, t_msg * 1e3, n / t_msg, t_pre * 1e3, n / t_pre, ok_msg, ok_pre, n), flush=True)
        for b in (d_q, d_m, d_s, d_ok, d_z, d_r, d_ss):
            b.free()
