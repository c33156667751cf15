"""-m gpu parity tests at BASELINE.json's full sizes, device-resident through the C ABI (ecgpu_dev_alloc + the *_dev
entry points; no torch): the 2^24-term k256 MSM as ONE plan (configs[3]), 2^20 p256 and p384 variable-base
multiplications (configs[2], configs[4]).  At these sizes the oracle cannot redo the whole job in seconds, so every
output is covered by a size-independent identity and the oracle checks strided samples bit for bit."""
import numpy as np
import pytest

import oracle_lib
import pyec
from gpu_common import ecgpu_module, scalars_to_int_sum

@pytest.fixture(scope="module")
def eng():
    e = ecgpu_module().Engine(0)
    oracle_lib.build()
    yield e
    e.close()


def fast_scalars(c, n, seed):
    """n uniformly random canonical scalars as an (n, L) uint8 array.  The orders of k256, p256 and p384 start with
    32 one bits, so clearing one bit of an all-ones top word (probability 2^-32) keeps every value below n."""
    assert c.n >> (8 * c.L - 32) == 0xFFFFFFFF
    b = np.random.default_rng(seed).integers(0, 256, (n, c.L), dtype=np.uint8)
    top = (b[:, 0] == 255) & (b[:, 1] == 255) & (b[:, 2] == 255) & (b[:, 3] == 255)
    b[top, 3] = 254
    return b


def dot_mod(k, s, mod):
    """sum_i k_i * s_i mod `mod` for two (n, L) big-endian byte arrays: 16-bit limbs of k against 8-bit limbs of s in
    float64 matrix products (every partial sum stays below 2^53, so the arithmetic is exact)."""
    n, L = k.shape
    total = 0
    step = 1 << 19
    for lo in range(0, n, step):
        kc = k[lo:lo + step].astype(np.float64)
        k16 = kc[:, 0::2] * 256.0 + kc[:, 1::2]                       # limb j has weight 2^(16 (L/2 - 1 - j))
        m = k16.T @ s[lo:lo + step].astype(np.float64)                 # (L/2, L), entries < 2^24 * 2^19
        for a in range(L // 2):
            wa = 16 * (L // 2 - 1 - a)
            for b in range(L):
                total += int(m[a, b]) << (wa + 8 * (L - 1 - b))
    return total % mod


def test_dot_mod_helper_is_exact():
    c = pyec.K256
    k, s = fast_scalars(c, 3000, 1), fast_scalars(c, 3000, 2)
    k[:7] = 255
    s[:7] = 255                                                       # worst-case limbs
    want = sum(int.from_bytes(bytes(k[i]), "big") * int.from_bytes(bytes(s[i]), "big") for i in range(3000)) % c.n
    assert dot_mod(k, s, c.n) == want


def _msm_dev(eng, cid, L, d_k, d_p, n, koff=0):
    d_o, d_f = eng.dev_alloc(256), eng.dev_alloc(16)
    eng.lincomb_dev(cid, d_k.at(koff * L), d_p.at(koff * 2 * L), None, n, d_o, d_f)
    out, inf = eng.to_host(d_o, 2 * L), int(eng.to_host(d_f, 1)[0])
    d_o.free(); d_f.free()
    return out, inf


@pytest.mark.gpu
def test_full_size_msm_k256_2p24_one_plan(eng):
    """configs[3]: a device-resident 2^24-term k256 MSM (one Pippenger plan: c = 16, two-level sort over 2^28 entries).
    (a) P_i = s_i G:  MSM == (sum k_i s_i mod n) G;   (b) MSM == MSM(first 2^23 terms) + MSM(last 2^23 terms);
    (c) all points = G:  MSM == (sum k_i) G;   (d) a strided 2^12-term sub-MSM against the oracle."""
    c = pyec.K256
    L, n = 32, 1 << 24
    k = fast_scalars(c, n, 0xEC000004)
    s = fast_scalars(c, n, 0xEC000054)
    k[0] = 0
    k[1, :] = np.frombuffer((c.n - 1).to_bytes(32, "big"), np.uint8)
    d_k, d_s = eng.to_device(k.reshape(-1)), eng.to_device(s.reshape(-1))
    d_p = eng.dev_alloc(n * 64)
    eng.mul_by_generator_dev(0, d_s, n, d_p, None)
    d_s.free()
    full, finf = _msm_dev(eng, 0, L, d_k, d_p, n)
    want, winf = oracle_lib.batch_mul_base(0, pyec.enc_scalar(c, dot_mod(k, s, c.n)))
    assert bytes(full) == bytes(want) and finf == int(winf[0]) == 0
    h = n // 2
    a, af = _msm_dev(eng, 0, L, d_k, d_p, h)
    b, bf = _msm_dev(eng, 0, L, d_k, d_p, h, koff=h)
    sm, sf = eng.point_sum(0, np.concatenate([a, b]), np.array([af, bf], np.uint8))
    assert bytes(sm) == bytes(full) and sf == finf
    # (d) every 4096th term: GPU sub-MSM == oracle (lincomb_vartime) on the same 4096 terms
    idx = np.arange(0, n, 4096)
    pts = eng.to_host(d_p).reshape(n, 64)
    sub_k, sub_p = k[idx].reshape(-1), pts[idx].reshape(-1)
    o, f = eng.lincomb(0, sub_k, sub_p)
    w, wf = oracle_lib.msm(0, sub_k, sub_p, vartime=True)
    assert bytes(o) == bytes(w) and f == wf
    # the points themselves: a strided sample of s_i G against the oracle
    w, _ = oracle_lib.batch_mul_base(0, s[idx[:64]].reshape(-1))
    assert bytes(pts[idx[:64]].reshape(-1)) == bytes(w)
    del pts
    # (c) all points = G
    gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
    eng.to_device(np.tile(gxy, n), d_p)
    o, f = _msm_dev(eng, 0, L, d_k, d_p, n)
    w, wf = oracle_lib.batch_mul_base(0, pyec.enc_scalar(c, scalars_to_int_sum(k.reshape(-1), 32, c.n)))
    assert bytes(o) == bytes(w) and f == int(wf[0])
    d_k.free(); d_p.free()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["p256", "p384"])
def test_full_size_variable_base_2p20(eng, curve):
    """configs[2] / configs[4]: 2^20 (scalar, point) pairs, device-resident.  k_i (s_i G) == (k_i s_i) G for EVERY
    element (variable-base kernel against the fixed-base kernel, two different algorithms), plus the oracle on the edge elements forced into the head of the batch
    and on a strided sample of 2^16 elements (every host core)."""
    c = pyec.CURVES[curve]
    L, n = c.L, 1 << 20
    k = fast_scalars(c, n, 0xEC000003 + c.cid)
    s = fast_scalars(c, n, 0xEC000053 + c.cid)
    for i, v in enumerate([0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 1 << 128]):
        k[i] = np.frombuffer(v.to_bytes(L, "big"), np.uint8)
    d_k, d_s = eng.to_device(k.reshape(-1)), eng.to_device(s.reshape(-1))
    d_p, d_o, d_f = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n * 2 * L), eng.dev_alloc(n + 16)
    eng.mul_by_generator_dev(c.cid, d_s, n, d_p, None)
    eng.mul_dev(c.cid, d_k, d_p, None, n, d_o, d_f)
    out, inf = eng.to_host(d_o), eng.to_host(d_f, n)
    kb, sb = bytes(k.reshape(-1)), bytes(s.reshape(-1))
    ks = b"".join((int.from_bytes(kb[L * i: L * i + L], "big") * int.from_bytes(sb[L * i: L * i + L], "big") % c.n).to_bytes(L, "big")
                  for i in range(n))
    eng.to_device(np.frombuffer(ks, np.uint8), d_s)
    eng.mul_by_generator_dev(c.cid, d_s, n, d_p, d_f)
    want, winf = eng.to_host(d_p), eng.to_host(d_f, n)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    assert inf[0] == 1 and inf[1:].sum() == 0                       # k_0 = 0 -> identity, nothing else
    # oracle, bit for bit: the forced edge elements and a strided sample
    eng.to_device(s.reshape(-1), d_s)
    eng.mul_by_generator_dev(c.cid, d_s, n, d_p, None)
    pts = eng.to_host(d_p).reshape(n, 2 * L)
    idx = np.concatenate([np.arange(8), np.arange(8, n, n // 56)])
    w, wf = oracle_lib.batch_mul(c.cid, k[idx].reshape(-1), pts[idx].reshape(-1))
    assert bytes(out.reshape(n, 2 * L)[idx].reshape(-1)) == bytes(w) and bytes(inf[idx]) == bytes(wf)
    # and one element in sixteen (2^16 of them) against the oracle's `ProjectivePoint * Scalar` on all host cores
    idx = np.arange(5, n, 16)
    w, wf = oracle_lib.batch_mul_mt(c.cid, k[idx].reshape(-1), pts[idx].reshape(-1))
    assert bytes(out.reshape(n, 2 * L)[idx].reshape(-1)) == bytes(w) and bytes(inf[idx]) == bytes(wf)
    for b in (d_k, d_s, d_p, d_o, d_f):
        b.free()
