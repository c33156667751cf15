#!/usr/bin/env python3
"""Randomised differential run on the GPU box: random parameter set (all twelve), sizes, window widths, chunk sizes, sort
modes, scalar split (GLV / plain), duplicated / cancelling / identity terms, random shard splits through the parts / finish
halves; every result against the oracle (n <= 2^13) or a group identity (larger n).  One-off evidence, not part of the test
suite:    FUZZ_SECONDS=200 FUZZ_SEED=1 python tools/gpu_fuzz.py"""
import importlib
import collections
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ec = importlib.import_module("elliptic-curves_amd")
import oracle_lib  # noqa: E402
import pyec  # noqa: E402
from gpu_common import rand_scalars, scalars_to_int_sum  # noqa: E402

oracle_lib.build()
e = ec.Engine(0, variant="knobs")   # the tool build: the ECGPU_* knobs below are read there only (csrc/ecgpu_knobs.h)
rng = random.Random(int(os.environ.get("FUZZ_SEED", "20260924")))
budget = float(os.environ.get("FUZZ_SECONDS", "150"))
t_end = time.time() + budget
stats = collections.defaultdict(int)
NAMES = ["k256", "p256", "p384", "sm2", "p224", "p192", "p521", "bp256", "bp384", "bp256t1", "bp384t1", "bign256"]
DEFAULT_W = {"k256": 26, "p256": 24, "p384": 20, "sm2": 24, "p224": 24, "p192": 24, "p521": 20, "bp256": 24, "bp384": 20,
             "bp256t1": 24, "bp384t1": 20, "bign256": 24}
pools = {}


def pool(c):
    if c.name not in pools:
        n = 1 << 16
        pts, _ = e.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xF0 + c.cid))
        pools[c.name] = pts.reshape(n, 2 * c.L).copy()
    return pools[c.name]


def int_sum(c, k):
    """sum of the scalars mod n (records in the curve's byte order)"""
    a = np.ascontiguousarray(k, np.uint8).reshape(-1, c.L)
    if c.le:
        a = a[:, ::-1]
    return scalars_to_int_sum(np.ascontiguousarray(a).reshape(-1), c.L, c.n)


def msm_knobs(c):
    os.environ["ECGPU_MSM_SORT2"] = rng.choice(["0", "1"])
    os.environ["ECGPU_MSM_SORT_PACKED"] = rng.choice(["1", "1", "0"])      # the packed two-level sort (round 4) / the round-3 kernels
    if rng.random() < 0.3:
        os.environ["ECGPU_MSM_CHUNK"] = str(rng.choice([1, 2, 3, 7, 33, 500, 100000]))
    else:
        os.environ.pop("ECGPU_MSM_CHUNK", None)
    os.environ["ECGPU_MSM_SMALL_LOG2"] = rng.choice(["-1", "16"])
    if c.name == "k256":
        os.environ["ECGPU_MSM_GLV"] = rng.choice(["0", "1", "auto"])
    else:
        os.environ.pop("ECGPU_MSM_GLV", None)


def note(exc_type, exc, tb):
    print("FUZZ CASE FAILED: kind=%s curve=%s n=%s knobs=%s" % (
        kind, c.name, n, {k_: v for k_, v in os.environ.items() if k_.startswith("ECGPU_")}), flush=True)
    sys.__excepthook__(exc_type, exc, tb)


sys.excepthook = note
kind = c = n = None
while time.time() < t_end:
    c = pyec.CURVES[rng.choice(NAMES)]
    L = c.L
    kind = rng.choice(["msm", "msm", "msm", "shards", "shard_lanes", "fixed", "var", "sig", "lanes"])
    if kind == "sig" and c.name == "bign256":
        kind = "bign"
    if kind == "sig" and c.name == "sm2":
        kind = "var"
    if kind == "msm":
        big = rng.random() < 0.25
        n = rng.randrange(1 << 14, 1 << 19) if big else rng.randrange(1, 1 << 13)
        cb = rng.choice([0, 0, rng.randrange(4, 17)])
        msm_knobs(c)
        e.set_msm_window(cb)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30)).copy().reshape(n, L)
        if big:
            gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
            o, f = e.lincomb(c.cid, k.reshape(-1), np.tile(gxy, n))
            w, wf = oracle_lib.batch_mul_base(c.cid, pyec.enc_scalar(c, int_sum(c, k)))
            assert bytes(o) == bytes(w) and f == int(wf[0]), ("msm property", c.name, n, cb, dict(os.environ))
            stats["msm_property"] += 1
        else:
            P = pool(c)
            idx = np.array([rng.randrange(1 << 16) for _ in range(n)])
            pts = P[idx].copy()
            inf = np.zeros(n, np.uint8)
            for _ in range(rng.randrange(0, 6)):           # duplicates, cancelling pairs, identities, tiny / huge scalars
                i, j = rng.randrange(n), rng.randrange(n)
                what = rng.randrange(5)
                if what == 0:
                    pts[j] = pts[i]; k[j] = k[i]; inf[j] = inf[i]
                elif what == 1:
                    pts[j] = pts[i]; inf[j] = inf[i]
                    k[j] = np.frombuffer(pyec.enc_scalar(c, (c.n - int.from_bytes(bytes(k[i]), c.order)) % c.n), np.uint8)
                elif what == 2:
                    inf[i] = 1; pts[i] = 0
                elif what == 3:
                    k[i] = np.frombuffer(pyec.enc_scalar(c, rng.choice([0, 1, 2, c.n - 1, c.n - 2])), np.uint8)
                else:
                    k[j] = k[i]
            o, f = e.lincomb(c.cid, k.reshape(-1), pts.reshape(-1), inf)
            w, wf = oracle_lib.msm(c.cid, k.reshape(-1), pts.reshape(-1), inf, vartime=True)
            assert bytes(o) == bytes(w) and f == wf, ("msm oracle", c.name, n, cb, dict(os.environ))
            stats["msm_oracle"] += 1
            if n <= 600 and rng.random() < 0.5:              # the constant-time `lincomb` on the same terms
                o2, f2 = e.lincomb_ct(c.cid, k.reshape(-1), pts.reshape(-1), inf)
                assert bytes(o2) == bytes(w) and f2 == wf, ("lincomb_ct", c.name, n)
                stats["lincomb_ct"] += 1
            if rng.random() < 0.4:                           # the same terms as x + SEC1 tag records (identities: tag 0, x = 0)
                ylow = pts[:, L] if c.le else pts[:, 2 * L - 1]
                tags = np.where(inf != 0, 0, 2 + (ylow & 1)).astype(np.uint8)
                xs = np.where(inf[:, None] != 0, 0, pts[:, :L]).astype(np.uint8)
                o3, f3 = e.lincomb_compressed(c.cid, k.reshape(-1), xs.reshape(-1), tags)
                assert bytes(o3) == bytes(w) and f3 == wf, ("msm_compressed", c.name, n, cb, dict(os.environ))
                stats["msm_compressed"] += 1
    elif kind == "lanes":
        # several MSMs in flight (ecgpu_set_msm_lanes on an asynchronous context) == the same MSMs one at a time
        msm_knobs(c)
        e.set_msm_window(0)
        P = pool(c)
        nj = rng.randrange(2, 7)
        jobs, want = [], []
        for _ in range(nj):
            n = rng.choice([rng.randrange(1, 200), rng.randrange(200, 1 << 13), rng.randrange(1 << 13, 1 << 17)])
            d_k = e.to_device(rand_scalars(c.cid, n, rng.randrange(1 << 30)))
            d_p = e.to_device(P[np.array([rng.randrange(1 << 16) for _ in range(n)])].copy().reshape(-1))
            d_o, d_f = e.dev_alloc(2 * L), e.dev_alloc(16)
            e.lincomb_dev(c.cid, d_k, d_p, None, n, d_o, d_f)
            want.append((bytes(e.to_host(d_o, 2 * L)), int(e.to_host(d_f, 1)[0])))
            e.to_device(np.zeros(2 * L, np.uint8), d_o)
            jobs.append((n, d_k, d_p, d_o, d_f))
        e.set_async(True)
        e.set_msm_lanes(rng.randrange(2, 5))
        for n, d_k, d_p, d_o, d_f in jobs:
            e.lincomb_dev(c.cid, d_k, d_p, None, n, d_o, d_f)
        e.synchronize()
        e.set_msm_lanes(1)
        e.set_async(False)
        for (n, d_k, d_p, d_o, d_f), w in zip(jobs, want):
            assert (bytes(e.to_host(d_o, 2 * L)), int(e.to_host(d_f, 1)[0])) == w, ("lanes", c.name, n, dict(os.environ))
            for b in (d_k, d_p, d_o, d_f):
                b.free()
        stats["msm_lanes"] += 1
    elif kind == "shards":
        # the two halves of a multi-GPU MSM on random, unequal shards (some empty) == the one-call MSM
        n = rng.randrange(1, 1 << 15)
        nsh = rng.randrange(1, 6)
        msm_knobs(c)
        e.set_msm_window(0)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        P = pool(c)
        pts = P[np.array([rng.randrange(1 << 16) for _ in range(n)])].copy().reshape(-1)
        cuts = sorted(rng.randrange(n + 1) for _ in range(nsh - 1))
        bounds = list(zip([0] + cuts, cuts + [n]))
        plan_terms = max(hi - lo for lo, hi in bounds) or 1
        want, wf = e.lincomb(c.cid, k, pts)
        nbytes = e.msm_parts_bytes(c.cid, plan_terms)
        d_all = e.dev_alloc(nsh * nbytes)
        for r_, (lo, hi) in enumerate(bounds):
            m = hi - lo
            dk = e.to_device(k[lo * L: hi * L]) if m else None
            dp = e.to_device(pts[lo * 2 * L: hi * 2 * L]) if m else None
            e.msm_parts_dev(c.cid, dk, dp, None, m, plan_terms, d_all.at(r_ * nbytes))
            for b in (dk, dp):
                if b is not None:
                    b.free()
        d_o, d_f = e.dev_alloc((2 * L + 15) // 16 * 16), e.dev_alloc(16)
        e.msm_finish_dev(c.cid, d_all, nsh, plan_terms, d_o, d_f)
        got, gf = e.to_host(d_o, 2 * L), int(e.to_host(d_f, 1)[0])
        assert bytes(got) == bytes(want) and gf == wf, ("shards", c.name, n, bounds, dict(os.environ))
        for b in (d_all, d_o, d_f):
            b.free()
        stats["msm_shards"] += 1
    elif kind == "shard_lanes":
        # sharded MSM steps in a software pipeline on rotating lanes (ecgpu_msm_parts_join_dev): several MSMs, each cut into random
        # shards whose local halves take the lanes in turn, the combining half of MSM i queued after the local halves of MSM i + 1 ==
        # the one-call MSMs (more local halves in flight than lanes: the implicit join when a lane comes around again)
        msm_knobs(c)
        e.set_msm_window(0)
        P = pool(c)
        nj, nsh = rng.randrange(2, 6), rng.randrange(1, 4)
        n = rng.choice([rng.randrange(1, 300), rng.randrange(300, 1 << 14), rng.randrange(1 << 14, 1 << 17)])
        cuts = sorted(rng.randrange(n + 1) for _ in range(nsh - 1))
        bounds = list(zip([0] + cuts, cuts + [n]))
        plan_terms = max(hi - lo for lo, hi in bounds) or 1
        pts = P[np.array([rng.randrange(1 << 16) for _ in range(n)])].copy().reshape(-1)
        ks = [rand_scalars(c.cid, n, rng.randrange(1 << 30)) for _ in range(nj)]
        want = [e.lincomb(c.cid, k, pts) for k in ks]
        nbytes = e.msm_parts_bytes(c.cid, plan_terms)
        lanes = rng.randrange(2, 5)
        nbuf = rng.randrange(2, 4)                           # gathered-record buffers in rotation (>= 2: the pipeline is two deep)
        d_p = [e.to_device(pts[lo * 2 * L: hi * 2 * L]) if hi > lo else None for lo, hi in bounds]
        d_k = [[e.to_device(k[lo * L: hi * L]) if hi > lo else None for lo, hi in bounds] for k in ks]
        d_all = [e.dev_alloc(nsh * nbytes) for _ in range(nbuf)]
        d_o = [e.dev_alloc((2 * L + 15) // 16 * 16 + 16) for _ in range(nj)]
        e.set_async(True)
        e.set_msm_lanes(lanes)
        pend = []

        def combine():
            i = pend.pop(0)
            buf = d_all[i % nbuf]
            for r_ in range(nsh):
                e.msm_parts_join_dev(buf.at(r_ * nbytes))
            e.msm_finish_dev(c.cid, buf, nsh, plan_terms, d_o[i].at(0), d_o[i].at((2 * L + 15) // 16 * 16))

        for i in range(nj):
            for r_, (lo, hi) in enumerate(bounds):
                e.msm_parts_dev(c.cid, d_k[i][r_], d_p[r_], None, hi - lo, plan_terms, d_all[i % nbuf].at(r_ * nbytes))
            pend.append(i)
            if len(pend) > 1:
                combine()
        while pend:
            combine()
        e.synchronize()
        e.set_msm_lanes(1)
        e.set_async(False)
        for i in range(nj):
            rec = e.to_host(d_o[i], (2 * L + 15) // 16 * 16 + 1)
            assert (bytes(rec[: 2 * L]), int(rec[(2 * L + 15) // 16 * 16])) == (bytes(want[i][0]), want[i][1]), \
                ("shard_lanes", c.name, n, bounds, lanes, nbuf, i, dict(os.environ))
        for b in d_all + d_o + [x for x in d_p if x is not None] + [x for row in d_k for x in row if x is not None]:
            b.free()
        stats["msm_shard_lanes"] += 1
    elif kind == "sig":
        # signatures made from engine-computed nonce points (valid ones), some disturbed: verification, public-key recovery
        # and message-level verification against the oracle, verdict for verdict and key for key
        import hashlib
        n = rng.randrange(1, 400)
        L = c.L
        ds = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        ks = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        Q, _ = e.mul_by_generator(c.cid, ds)
        R, _ = e.mul_by_generator(c.cid, ks)
        H = {"k256": "sha256", "p256": "sha256", "p384": "sha384", "p224": "sha224", "p521": "sha512", "bp256": "sha256", "bp384": "sha384",
             "bp256t1": "sha256", "bp384t1": "sha384"}.get(c.name)
        msg_len = rng.choice([0, 1, 31, 55, 56, 64, 111, 112, 127, 128, 129, 300])
        msgs = bytes(rng.randrange(256) for _ in range(n * msg_len))
        zs, rr, ss, ids = bytearray(), bytearray(), bytearray(), bytearray()
        for i in range(n):
            d = int.from_bytes(bytes(ds[i * L:(i + 1) * L]), "big")
            k = int.from_bytes(bytes(ks[i * L:(i + 1) * L]), "big") or 1
            if H:
                dg = hashlib.new(H, msgs[i * msg_len:(i + 1) * msg_len]).digest()
                zb = dg[:L] if len(dg) >= L else bytes(L - len(dg)) + dg
            else:
                zb = bytes(rng.randrange(256) for _ in range(L))
                if c.n.bit_length() < 8 * L:
                    zb = b"\0" + zb[1:]
            zi = int.from_bytes(zb, "big")
            x = int.from_bytes(bytes(R[2 * L * i: 2 * L * i + L]), "big")
            ri = x % c.n
            si = pow(k, -1, c.n) * (zi + ri * d) % c.n
            flip = rng.randrange(6)
            if flip == 0:
                si = (si + 1) % c.n
            if flip == 1:
                ri = (ri + 1) % c.n
            zs += zb; rr += ri.to_bytes(L, "big"); ss += si.to_bytes(L, "big")
            ids.append((int(R[2 * L * i + 2 * L - 1]) & 1) ^ (1 if flip == 2 else 0) | (2 if x >= c.n or flip == 3 else 0))
        high = bool(rng.randrange(2))
        v = e.ecdsa_verify(c.cid, bytes(zs), bytes(rr), bytes(ss), Q, reject_high_s=high)
        assert bytes(v) == bytes(oracle_lib.ecdsa_verify(c.cid, bytes(zs), bytes(rr), bytes(ss), Q, reject_high_s=high)), ("verify", c.name, n)
        stats["verify"] += 1
        if True:                                             # (p224 included since its square root exists: round 3)
            gk, gv = e.ecdsa_recover(c.cid, bytes(zs), bytes(rr), bytes(ss), bytes(ids), reject_high_s=high)
            wk, wv = oracle_lib.ecdsa_recover(c.cid, bytes(zs), bytes(rr), bytes(ss), bytes(ids), reject_high_s=high)
            assert bytes(gk) == bytes(wk) and bytes(gv) == bytes(wv), ("recover", c.name, n)
            stats["recover"] += 1
        if H:
            sg = b"".join(bytes(rr[i * L:(i + 1) * L]) + bytes(ss[i * L:(i + 1) * L]) for i in range(n))
            vm = e.ecdsa_verify_msg(c.cid, Q, msgs, msg_len, sg, reject_high_s=high)
            assert bytes(vm) == bytes(v) == bytes(oracle_lib.ecdsa_verify_msg(c.cid, Q, msgs, msg_len, sg, reject_high_s=high)), ("verify_msg", c.name, n, msg_len)
            stats["verify_msg"] += 1
    elif kind == "bign":
        # bign signatures over engine-computed keys and nonce points, a third of them disturbed somewhere: verification on the
        # prehash and from the messages (belt-hash on the device) against the oracle, verdict for verdict
        n = rng.randrange(1, 300)
        ds = bytes(rand_scalars(c.cid, n, rng.randrange(1 << 30)))
        ks = bytes(rand_scalars(c.cid, n, rng.randrange(1 << 30)))
        Q = bytes(e.mul_by_generator(c.cid, ds)[0])
        R = bytes(e.mul_by_generator(c.cid, ks)[0])
        msg_len = rng.choice([0, 1, 13, 31, 32, 33, 64, 65, 200])
        msgs = bytes(rng.randrange(256) for _ in range(n * msg_len))
        hs, sigs = bytearray(), bytearray()
        for i in range(n):
            h = oracle_lib.belt_hash(msgs[i * msg_len:(i + 1) * msg_len])
            s0 = oracle_lib.belt_hash(pyec.BELT_OID + R[64 * i:64 * i + 32] + h)[:16]
            d, k = int.from_bytes(ds[32 * i:32 * i + 32], "little"), int.from_bytes(ks[32 * i:32 * i + 32], "little")
            s1 = (k - int.from_bytes(h, "little") - (int.from_bytes(s0, "little") + 2 ** 128) * d) % c.n
            sig = bytearray(s0 + s1.to_bytes(32, "little"))
            flip = rng.randrange(9)
            if flip == 0:
                sig[rng.randrange(48)] ^= 1 << rng.randrange(8)
            elif flip == 1:
                sig[16:] = rng.choice([0, c.n, c.n + 1, (1 << 256) - 1]).to_bytes(32, "little")
            elif flip == 2:
                sig[:16] = bytes(16)
            hs += h
            sigs += sig
        v = e.bign_verify(bytes(hs), bytes(sigs), Q)
        assert bytes(v) == bytes(oracle_lib.bign_verify(bytes(hs), bytes(sigs), Q)), ("bign", n)
        vm = e.bign_verify_msg(Q, msgs, msg_len, bytes(sigs))
        assert bytes(vm) == bytes(v) == bytes(oracle_lib.bign_verify_msg(Q, msgs, msg_len, bytes(sigs))), ("bign_msg", n, msg_len)
        assert 0 < int(v.sum()) or n < 12        # (a sanity check of the CASE, not of the engine: a third of the signatures are disturbed, so twelve disturbed ones in a row are a 2e-6 event)
        stats["bign_verify"] += 1
    elif kind == "fixed":
        n = rng.randrange(1, 3000)
        wdt = rng.choice([0, 0, rng.randrange(4, 17)])
        if wdt:
            e.set_base_window(c.cid, wdt)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        ct = rng.random() < 0.3                              # the uniform-schedule entry point instead
        o, f = e.mul_by_generator(c.cid, k, constant_time=ct)
        w, wf = oracle_lib.batch_mul_base(c.cid, k)
        assert bytes(o) == bytes(w) and bytes(f) == bytes(wf), ("fixed", c.name, n, wdt, ct)
        stats["fixed_ct"] += 1 if ct else 0
        if wdt:
            e.set_base_window(c.cid, 0)                       # un-pinned: back to the table policy
        stats["fixed"] += 1
    else:
        n = rng.randrange(1, 1500)
        k = rand_scalars(c.cid, n, rng.randrange(1 << 30))
        P = pool(c)
        pts = P[np.array([rng.randrange(1 << 16) for _ in range(n)])].copy().reshape(-1)
        ct = rng.random() < 0.4
        inf = None
        if rng.random() < 0.3:                               # identities among the points, tiny / huge scalars
            inf = np.zeros(n, np.uint8)
            k = k.copy()
            for _ in range(rng.randrange(1, 5)):
                i = rng.randrange(n)
                if rng.random() < 0.5:
                    inf[i] = 1
                else:
                    k[i * L:(i + 1) * L] = np.frombuffer(pyec.enc_scalar(c, rng.choice([0, 1, 2, c.n - 1, c.n - 2])), np.uint8)
        o, f = e.mul(c.cid, k, pts, inf, constant_time=ct)
        w, wf = oracle_lib.batch_mul(c.cid, k, pts, inf)
        assert bytes(o) == bytes(w) and bytes(f) == bytes(wf), ("var", c.name, n, ct)
        stats["var_ct" if ct else "var"] += 1
        if rng.random() < 0.3:                               # decompression of the x-coordinates just computed (and of junk)
            xs = o.reshape(n, 2 * L)[:, :L].copy()
            odd = np.array([rng.randrange(2) for _ in range(n)], np.uint8)
            for _ in range(rng.randrange(0, 4)):
                xs[rng.randrange(n)] = np.frombuffer(bytes(rng.randrange(256) for _ in range(L)), np.uint8)
            dxy, dok = e.decompress(c.cid, xs.reshape(-1), odd)
            wxy, wok = oracle_lib.batch_decompress(c.cid, xs.reshape(-1), odd)
            assert bytes(dxy) == bytes(wxy) and bytes(dok) == bytes(wok), ("decompress", c.name, n)
            stats["decompress"] += 1
print("fuzz ok: %s in %.0f s, seed %s" % (dict(sorted(stats.items())), budget, os.environ.get("FUZZ_SEED", "20260924")))
