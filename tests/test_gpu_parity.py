"""-m gpu parity tests: the HIP path, called through the C ABI (ctypes -> libecgpu.so), compared
bit-for-bit with (1) the reference's golden vectors in tests/golden/, (2) the oracle on the same
seeded inputs, and (3) at BASELINE sizes, size-independent group identities."""
import os

import numpy as np
import pytest

import oracle_lib
import pyec
from gpu_common import (ALL_CURVES, comb_corner_scalars, msm_exceptional_terms, CURVES, ecdsa_cases, ecdsa_pack, ecgpu_module, schnorr_inputs, edge_scalars, ladder_edge_scalars, load_golden,
                        rand_scalars, scalars_to_int_sum)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    ecgpu = ecgpu_module()
    e = ecgpu.Engine(0)          # raises without the HIP extension / a gfx950 device: no fallback
    yield e
    e.close()


@pytest.fixture(scope="module")
def keng():
    """An engine on the TOOL build of the library (lib/libecgpu_knobs.so, csrc/ecgpu_knobs.h): the same kernel objects, but the ECGPU_*
    tuning knobs are read from the environment — the product library never reads it.  The tests that force a code path the planner
    would not pick at their sizes (two-level sort at small n, chunk sizes, the chunked host-pointer MSM, the fused tail) run on it."""
    ecgpu = ecgpu_module()
    e = ecgpu.Engine(0, variant="knobs")
    yield e
    e.close()


@pytest.fixture(scope="module", autouse=True)
def _oracle_built():
    oracle_lib.build()


def xy(v):
    return bytes.fromhex(v["x"]) + bytes.fromhex(v["y"])


# ---------------------------------------------------------------------------------------------------
# golden vectors of the reference
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("curve", CURVES)
def test_golden_group_vectors(eng, curve):
    """{k256,p256,p384}/src/test_vectors/group.rs through mul_by_generator, mul and lincomb(1 term)."""
    c = pyec.CURVES[curve]
    g = load_golden(curve)["group"]
    ks = [pyec.enc_scalar(c, v["k"]) for v in g["add"]] + [bytes.fromhex(v["k"]) for v in g["mul"]]
    want = b"".join(xy(v) for v in g["add"] + g["mul"])
    scal = b"".join(ks)
    n = len(ks)
    gxy = pyec.enc_point(c, pyec.G(c))[0]
    out, inf = eng.mul_by_generator(c.cid, scal)
    assert bytes(out) == want and not inf.any()
    out, inf = eng.mul(c.cid, scal, gxy * n)
    assert bytes(out) == want and not inf.any()
    for i in (0, 1, 19, 20, n - 1):
        o, f = eng.lincomb(c.cid, ks[i], gxy)
        assert bytes(o) == want[2 * c.L * i: 2 * c.L * (i + 1)] and f == 0


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_var_base_ladder_corner_cases(eng, oracle, curve):
    """Scalars that put the incomplete Jacobian ladder at its limits (ecgpu_varmul.h): accumulator equal to
    +-(table operand) at the last digit, late start, digit -8 runs, carry into the top digit."""
    c = pyec.CURVES[curve]
    ks = ladder_edge_scalars(c)
    rng = np.random.default_rng(0x1ADDE5 + c.cid)
    scal = b"".join(pyec.enc_scalar(c, k) for k in ks)
    for P in (pyec.G(c), pyec.mul(c, int(rng.integers(2, 2 ** 62)), pyec.G(c))):
        pxy = pyec.enc_point(c, P)[0] * len(ks)
        out, inf = eng.mul(c.cid, scal, pxy)
        want, winf = oracle.batch_mul(c.cid, scal, pxy, None)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)


@pytest.mark.parametrize("curve", CURVES)
def test_golden_add_vectors_via_point_sum(eng, curve):
    """ADD_TEST_VECTORS: k*G as a sum of k copies of G (complete addition incl. the doubling case)."""
    c = pyec.CURVES[curve]
    g = load_golden(curve)["group"]["add"]
    gxy = pyec.enc_point(c, pyec.G(c))[0]
    for v in g:
        o, f = eng.point_sum(c.cid, gxy * v["k"])
        assert bytes(o) == xy(v) and f == 0


@pytest.mark.parametrize("curve", CURVES)
def test_golden_ecdsa_vectors(eng, curve):
    """{p256,p384,k256}/src/test_vectors/ecdsa.rs: d*G == Q and x(k*G) == r via the fixed-base kernel;
    x(u1*G + u2*Q) == r via mul_by_generator_and_mul_add (verification shape) and via a 2-term lincomb."""
    c = pyec.CURVES[curve]
    vec = load_golden(curve)["ecdsa"]
    scal = b"".join(bytes.fromhex(v["d"]) + bytes.fromhex(v["k"]) for v in vec)
    out, inf = eng.mul_by_generator(c.cid, scal)
    a_s, b_s, qs = b"", b"", b""
    for i, v in enumerate(vec):
        q = bytes(out[4 * c.L * i: 4 * c.L * i + 2 * c.L])
        assert q == bytes.fromhex(v["q_x"]) + bytes.fromhex(v["q_y"])
        rx = int.from_bytes(bytes(out[4 * c.L * i + 2 * c.L: 4 * c.L * i + 3 * c.L]), "big")
        r, s, z = int(v["r"], 16), int(v["s"], 16), int(v["m"], 16)
        assert rx % c.n == r
        w = pow(s, -1, c.n)
        a_s += pyec.enc_scalar(c, z * w % c.n)
        b_s += pyec.enc_scalar(c, r * w % c.n)
        qs += q
    res, rinf = eng.mul_by_generator_and_mul_add(c.cid, a_s, b_s, qs)
    gxy = pyec.enc_point(c, pyec.G(c))[0]
    for i, v in enumerate(vec):
        assert rinf[i] == 0
        assert int.from_bytes(bytes(res[2 * c.L * i: 2 * c.L * i + c.L]), "big") % c.n == int(v["r"], 16)
        o, f = eng.lincomb(c.cid, a_s[c.L * i: c.L * (i + 1)] + b_s[c.L * i: c.L * (i + 1)],
                           gxy + qs[2 * c.L * i: 2 * c.L * (i + 1)])
        assert bytes(o) == bytes(res[2 * c.L * i: 2 * c.L * (i + 1)]) and f == 0


# ---------------------------------------------------------------------------------------------------
# oracle parity on seeded inputs
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("curve", ALL_CURVES)
@pytest.mark.parametrize("window", [24, 16, 15, 13, 5, 4])
def test_fixed_base_vs_oracle(eng, curve, window):
    c = pyec.CURVES[curve]
    eng.set_base_window(c.cid, window)
    n = 3000 if window >= 16 else 700
    scal = rand_scalars(c.cid, n, 0xEC000002 + c.cid)
    # the comb's Jacobian additions are incomplete: corner scalars drive them towards accumulator = +-entry
    edge = b"".join(pyec.enc_scalar(c, k) for k in edge_scalars(c) + comb_corner_scalars(c, window))
    scal = np.concatenate([np.frombuffer(edge, np.uint8), scal])
    out, inf = eng.mul_by_generator(c.cid, scal)
    want, winf = oracle_lib.batch_mul_base(c.cid, scal)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    assert inf[0] == 1 and not out[: 2 * c.L].any()          # k = 0 -> identity encoding
    eng.set_base_window(c.cid, 0)        # un-pinned: back to the table policy


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_fixed_base_default_window_corner_scalars(eng, curve):
    c = pyec.CURVES[curve]
    w = {"k256": 26, "p256": 24, "p384": 20, "sm2": 24, "p224": 24, "p192": 24, "p521": 20, "bp256": 24, "bp384": 20, "bp256t1": 24, "bp384t1": 20}[curve]
    scal = b"".join(pyec.enc_scalar(c, k) for k in comb_corner_scalars(c, w) + edge_scalars(c))
    out, inf = eng.mul_by_generator(c.cid, scal)
    want, winf = oracle_lib.batch_mul_base(c.cid, scal)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_variable_base_vs_oracle(eng, curve):
    c = pyec.CURVES[curve]
    n = 600
    pts_s = rand_scalars(c.cid, n, 0xEC000013 + c.cid)
    pts, _ = oracle_lib.batch_mul_base(c.cid, pts_s)
    pts = pts.copy()
    scal = rand_scalars(c.cid, n, 0xEC000003 + c.cid).copy()
    inf = np.zeros(n, np.uint8)
    G = pyec.G(c)
    ek = edge_scalars(c)
    slot = 0
    for k in ek:                      # edge scalars x {G, -G, identity}
        for P in (G, pyec.neg(c, G), pyec.INF):
            e, f = pyec.enc_point(c, P)
            scal[slot * c.L: (slot + 1) * c.L] = np.frombuffer(pyec.enc_scalar(c, k), np.uint8)
            pts[slot * 2 * c.L: (slot + 1) * 2 * c.L] = np.frombuffer(e, np.uint8)
            inf[slot] = f
            slot += 1
    pts[(slot) * 2 * c.L: (slot + 1) * 2 * c.L] = pts[(slot + 1) * 2 * c.L: (slot + 2) * 2 * c.L]   # duplicate point
    out, oinf = eng.mul(c.cid, scal, pts, inf)
    want, winf = oracle_lib.batch_mul(c.cid, scal, pts, inf)
    assert bytes(out) == bytes(want) and bytes(oinf) == bytes(winf)
    want2, winf2 = oracle_lib.batch_mul(c.cid, scal[: 40 * c.L], pts[: 80 * c.L], inf[:40], vartime=True)
    assert bytes(out[: 80 * c.L]) == bytes(want2)
    # NULL flags path
    out3, _ = eng.mul(c.cid, scal[slot * c.L:], pts[slot * 2 * c.L:], None)
    assert bytes(out3) == bytes(want[slot * 2 * c.L:])


@pytest.mark.parametrize("curve", ALL_CURVES)
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 257, 4000])
def test_msm_vs_oracle(eng, keng, curve, n, monkeypatch):
    c = pyec.CURVES[curve]
    if n == 0:
        o, f = eng.lincomb(c.cid, b"", b"")
        assert f == 1 and not o.any()
        return
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xEC000014 + c.cid + n))
    pts = pts.copy()
    scal = rand_scalars(c.cid, n, 0xEC000004 + c.cid + n).copy()
    inf = np.zeros(n, np.uint8)
    if n >= 17:
        ek = edge_scalars(c)
        for i, k in enumerate(ek[:10]):
            scal[i * c.L: (i + 1) * c.L] = np.frombuffer(pyec.enc_scalar(c, k), np.uint8)
        inf[11] = 1; pts[11 * 2 * c.L: 12 * 2 * c.L] = 0                                   # identity term
        pts[13 * 2 * c.L: 14 * 2 * c.L] = pts[12 * 2 * c.L: 13 * 2 * c.L]                    # duplicate points
        neg = pyec.neg(c, pyec.dec_point(c, bytes(pts[14 * 2 * c.L: 15 * 2 * c.L]), 0))     # P_15 = -P_14, same scalar
        pts[15 * 2 * c.L: 16 * 2 * c.L] = np.frombuffer(pyec.enc_point(c, neg)[0], np.uint8)
        scal[15 * c.L: 16 * c.L] = scal[14 * c.L: 15 * c.L]
    want, winf = oracle_lib.msm(c.cid, scal, pts, inf, vartime=True)
    for cbits in ((0, 5) if n < 257 else (0, 4, 9, 12, 16)):           # the product library: its own plan for every window width
        eng.set_msm_window(cbits)
        o, f = eng.lincomb(c.cid, scal, pts, inf)
        assert bytes(o) == bytes(want) and f == winf, (curve, n, cbits)
    eng.set_msm_window(0)
    for sort2 in ("0", "1"):                  # single-level / two-level (partition, then buckets) counting sort, forced (tool build)
        monkeypatch.setenv("ECGPU_MSM_SORT2", sort2)
        for cbits in ((0, 5) if n < 257 else (0, 4, 9, 12, 16)):       # 0: automatic (small n: per-term products + tree sum)
            keng.set_msm_window(cbits)
            o, f = keng.lincomb(c.cid, scal, pts, inf)
            assert bytes(o) == bytes(want) and f == winf, (curve, n, cbits, sort2)
    monkeypatch.delenv("ECGPU_MSM_SORT2")
    keng.set_msm_window(0)
    if n == 17:
        want_ct, wf = oracle_lib.msm(c.cid, scal, pts, inf, vartime=False)
        assert bytes(want_ct) == bytes(want) and wf == winf


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_msm_chunk_sizes_and_skewed_scalars(eng, keng, curve, monkeypatch):
    """The accumulation lanes own fixed-size chunks of the sorted run, not buckets.  Sweep the chunk size (1 entry
    per lane .. everything in one lane) against the oracle, then feed scalar sets that put every term of a window
    into ONE bucket (all scalars equal / all ones): sum_i k P_i = k * sum_i P_i."""
    import time
    c = pyec.CURVES[curve]
    n = 3000
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xEC000015 + c.cid))
    scal = rand_scalars(c.cid, n, 0xEC000005 + c.cid)
    want, winf = oracle_lib.msm(c.cid, scal, pts, None, vartime=True)
    for chunk in ("1", "33", "100000"):
        monkeypatch.setenv("ECGPU_MSM_CHUNK", chunk)
        for cbits in (0, 9):
            keng.set_msm_window(cbits)
            o, f = keng.lincomb(c.cid, scal, pts)
            assert bytes(o) == bytes(want) and f == winf, (chunk, cbits)
    monkeypatch.delenv("ECGPU_MSM_CHUNK")
    keng.set_msm_window(0)
    n = 1 << 18
    pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xEC000016 + c.cid))
    total, tf = eng.point_sum(c.cid, pts)
    assert tf == 0
    k0 = bytes(rand_scalars(c.cid, 1, 0xEC000017 + c.cid))
    for k, sort2 in ((k0, "0"), (k0, "1"), (pyec.enc_scalar(c, 1), "1"), (pyec.enc_scalar(c, c.n - 1), "0")):
        monkeypatch.setenv("ECGPU_MSM_SORT2", sort2)          # "1": the two-level sort, whatever n is
        t0 = time.time()
        o, f = keng.lincomb(c.cid, np.tile(np.frombuffer(k, np.uint8), n), pts)
        dt = time.time() - t0
        w, wf = eng.mul(c.cid, k, total)
        assert bytes(o) == bytes(w) and f == int(wf[0])
        assert dt < 5.0, "skewed MSM took %.1f s: one lane is walking a whole bucket" % dt
    monkeypatch.delenv("ECGPU_MSM_SORT2")


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_msm_exceptional_additions(eng, keng, curve, monkeypatch):
    """Bucket runs with duplicates, P + Q next to P and Q, cancelling sums: the stretches whose incomplete XYZZ sum
    fails the exactness test are redone with the complete formulas (ecgpu_msm_chunk.h)."""
    import random
    c = pyec.CURVES[curve]
    ks, pts = msm_exceptional_terms(c, random.Random(0xE8CF + c.cid), filler=300)
    scal = b"".join(pyec.enc_scalar(c, k) for k in ks)
    pxy = b"".join(pyec.enc_point(c, P)[0] for P in pts)
    want, winf = oracle_lib.msm(c.cid, scal, pxy, None, vartime=True)
    for chunk in (None, "1", "2", "3", "40"):
        if chunk is None:
            monkeypatch.delenv("ECGPU_MSM_CHUNK", raising=False)
        else:
            monkeypatch.setenv("ECGPU_MSM_CHUNK", chunk)
        for cbits in (0, 4, 9):
            keng.set_msm_window(cbits)
            o, f = keng.lincomb(c.cid, scal, pxy)
            assert bytes(o) == bytes(want) and f == winf, (chunk, cbits)
    monkeypatch.delenv("ECGPU_MSM_CHUNK", raising=False)
    keng.set_msm_window(0)
    # a whole run of one repeated term: every stretch takes the complete path
    n = 5000
    o, f = keng.lincomb(c.cid, scal[: c.L] * n, pxy[: 2 * c.L] * n)
    w, wf = eng.mul(c.cid, pyec.enc_scalar(c, ks[0] * n % c.n), pxy[: 2 * c.L])
    assert bytes(o) == bytes(w) and f == int(wf[0])


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_msm_cancels_to_identity(eng, curve):
    c = pyec.CURVES[curve]
    n = 64
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 77))
    scal = rand_scalars(c.cid, n, 78)
    neg_scal = b"".join(pyec.enc_scalar(c, (c.n - int.from_bytes(bytes(scal[i * c.L: (i + 1) * c.L]), "big")) % c.n) for i in range(n))
    o, f = eng.lincomb(c.cid, np.concatenate([scal, np.frombuffer(neg_scal, np.uint8)]), np.concatenate([pts, pts]))
    assert f == 1 and not o.any()


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_batch_normalize_and_point_sum_vs_oracle(eng, curve):
    c = pyec.CURVES[curve]
    rng = np.random.default_rng(5 + c.cid)
    n = 300
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xB0))
    xyz = bytearray()
    for i in range(n):
        z = int.from_bytes(rng.bytes(c.L), "big") % c.p or 1
        if i % 50 == 7:
            xyz += (0).to_bytes(c.L, "big") + (1).to_bytes(c.L, "big") + (0).to_bytes(c.L, "big")
            continue
        x = int.from_bytes(bytes(pts[2 * c.L * i: 2 * c.L * i + c.L]), "big")
        y = int.from_bytes(bytes(pts[2 * c.L * i + c.L: 2 * c.L * (i + 1)]), "big")
        xyz += (x * z % c.p).to_bytes(c.L, "big") + (y * z % c.p).to_bytes(c.L, "big") + z.to_bytes(c.L, "big")
    out, inf = eng.batch_normalize(c.cid, bytes(xyz))
    want, winf = oracle_lib.batch_normalize(c.cid, bytes(xyz))
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf) and inf[7] == 1
    ones = np.tile(np.array([0] * (c.L - 1) + [1], np.uint8), n)
    o, f = eng.point_sum(c.cid, pts)
    w, wf = oracle_lib.msm(c.cid, ones, pts)
    assert bytes(o) == bytes(w) and f == wf
    o, f = eng.point_sum(c.cid, b"")
    assert f == 1


def test_k256_glv_decompose_vs_oracle(eng):
    c = pyec.K256
    ks = edge_scalars(c)
    scal = np.concatenate([np.frombuffer(b"".join(pyec.enc_scalar(c, k) for k in ks), np.uint8), rand_scalars(0, 5000, 0x61)])
    r1, r2 = eng.k256_glv_decompose(scal)
    for i in range(scal.size // 32):
        w1, w2 = oracle_lib.k256_glv_decompose(bytes(scal[32 * i: 32 * i + 32]))
        assert bytes(r1[32 * i: 32 * i + 32]) == w1 and bytes(r2[32 * i: 32 * i + 32]) == w2


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_error_behaviour(eng, curve):
    """Decoding errors mirror the reference: scalar >= n and off-curve / out-of-range points are refused."""
    ecgpu = ecgpu_module()
    c = pyec.CURVES[curve]
    good_k = pyec.enc_scalar(c, 5)
    gxy = pyec.enc_point(c, pyec.G(c))[0]
    for bad_k in (c.n, c.n + 1, 2 ** (8 * c.L) - 1):
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.mul_by_generator(c.cid, good_k + bad_k.to_bytes(c.L, "big"))
        assert e.value.code == ecgpu.ERR_SCALAR_RANGE
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.lincomb(c.cid, good_k + bad_k.to_bytes(c.L, "big"), gxy * 2)
        assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    off = bytearray(gxy); off[-1] ^= 1
    big = c.p.to_bytes(c.L, "big") + gxy[c.L:]
    for bad_p in (bytes(off), big):
        for fn in (eng.mul, eng.lincomb):
            with pytest.raises(ecgpu.EcgpuError) as e:
                fn(c.cid, good_k * 2, gxy + bad_p)
            assert e.value.code == ecgpu.ERR_POINT
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.mul_by_generator(99, good_k)
    assert e.value.code == ecgpu.ERR_CURVE
    out, inf = eng.mul_by_generator(c.cid, b"")          # empty batch
    assert out.size == 0


# ---------------------------------------------------------------------------------------------------
# BASELINE sizes: size-independent identities
# ---------------------------------------------------------------------------------------------------

def test_full_size_fixed_base_k256(eng):
    """config 2: 2^20 random k256 scalars.  (a) EVERY output against the oracle, byte for byte (the oracle's `mul_by_generator` on all
    host cores: a second or two); (b) checksum of checksums: sum_i (k_i G) == (sum_i k_i) G with the left side summed on the GPU
    over all 2^20 outputs; (c) the constant-time form (ecgpu_batch_mul_base_ct) returns the same bytes."""
    c = pyec.K256
    n = 1 << 20
    scal = rand_scalars(0, n, 0xEC000002)
    out, inf = eng.mul_by_generator(0, scal)
    assert not inf.any()
    want, winf = oracle_lib.batch_mul_base_mt(0, scal)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    out_ct, inf_ct = eng.mul_by_generator(0, scal, constant_time=True)
    assert bytes(out_ct) == bytes(want) and bytes(inf_ct) == bytes(winf)
    total = scalars_to_int_sum(scal, 32, c.n)
    o, f = eng.point_sum(0, out)
    w, wf = oracle_lib.batch_mul_base(0, pyec.enc_scalar(c, total))
    assert bytes(o) == bytes(w) and f == int(wf[0])


def test_full_size_variable_base_p256_sample(eng):
    """config 3 shape (p256 ECDH), 2^16 pairs here (the 2^20 run is bench.py --workload var_p256):
    linearity  k*(s*G) == (k*s)*G  for every element, checked through the fixed-base kernel, plus an
    oracle sample."""
    c = pyec.P256
    n = 1 << 16
    s = rand_scalars(1, n, 0xEC000023)
    k = rand_scalars(1, n, 0xEC000003)
    pts, _ = eng.mul_by_generator(1, s)
    out, inf = eng.mul(1, k, pts)
    ks = b"".join(pyec.enc_scalar(c, int.from_bytes(bytes(k[32 * i: 32 * i + 32]), "big") * int.from_bytes(bytes(s[32 * i: 32 * i + 32]), "big") % c.n)
                  for i in range(n))
    want, winf = eng.mul_by_generator(1, ks)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    wo, _ = oracle_lib.batch_mul(1, k[: 32 * 64], pts[: 64 * 64])
    assert bytes(out[: 64 * 64]) == bytes(wo)


@pytest.mark.parametrize("curve", ["k256", "p256"])
def test_msm_multi_tile_ragged(eng, keng, curve, monkeypatch):
    """More than one counting-sort tile with a ragged tail (n = 2^19 + 12345), identities sprinkled in, both
    the automatic window and c = 16: split linearity + all-G checksum."""
    c = pyec.CURVES[curve]
    n = (1 << 19) + 12345
    k = rand_scalars(c.cid, n, 0xEC0000A4 + c.cid)
    s = rand_scalars(c.cid, n, 0xEC0000B4 + c.cid)
    pts, _ = eng.mul_by_generator(c.cid, s)
    pts = pts.copy()
    inf = np.zeros(n, np.uint8)
    inf[::1001] = 1
    pts.reshape(n, 2 * c.L)[::1001] = 0
    h = (1 << 19) - 7
    for cbits, sort2 in ((0, "0"), (16, "0"), (16, "1"), (11, "1")):
        monkeypatch.setenv("ECGPU_MSM_SORT2", sort2)
        keng.set_msm_window(cbits)
        full, ff = keng.lincomb(c.cid, k, pts, inf)
        a, af = keng.lincomb(c.cid, k[: c.L * h], pts[: 2 * c.L * h], inf[:h])
        b, bf = keng.lincomb(c.cid, k[c.L * h:], pts[2 * c.L * h:], inf[h:])
        sm, sf = eng.point_sum(c.cid, np.concatenate([a, b]), np.array([af, bf], np.uint8))
        assert bytes(sm) == bytes(full) and sf == ff
        gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
        o, f = keng.lincomb(c.cid, k, np.tile(gxy, n))
        w, wf = oracle_lib.batch_mul_base(c.cid, pyec.enc_scalar(c, scalars_to_int_sum(k, c.L, c.n)))
        assert bytes(o) == bytes(w) and f == int(wf[0])
    monkeypatch.delenv("ECGPU_MSM_SORT2")
    keng.set_msm_window(0)
    # oracle on a sample of the same data
    m = 3000
    o, f = keng.lincomb(c.cid, k[: c.L * m], pts[: 2 * c.L * m], inf[:m])
    w, wf = oracle_lib.msm(c.cid, k[: c.L * m], pts[: 2 * c.L * m], inf[:m], vartime=True)
    assert bytes(o) == bytes(w) and f == wf


def test_full_size_msm_k256_properties(eng):
    """config 4 shape at 2^20 terms (2^24 is exercised by bench.py --workload msm_k256 --check):
    (a) all points = G: MSM == (sum k_i) G;  (b) distinct points P_i = s_i G: MSM == (sum k_i s_i) G;
    (c) split linearity MSM(all) == MSM(first half) + MSM(second half);  (d) permutation invariance."""
    c = pyec.K256
    n = 1 << 20
    k = rand_scalars(0, n, 0xEC000004)
    gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
    o, f = eng.lincomb(0, k, np.tile(gxy, n))
    w, wf = oracle_lib.batch_mul_base(0, pyec.enc_scalar(c, scalars_to_int_sum(k, 32, c.n)))
    assert bytes(o) == bytes(w) and f == int(wf[0])
    s = rand_scalars(0, n, 0xEC000024)
    pts, _ = eng.mul_by_generator(0, s)
    full, ff = eng.lincomb(0, k, pts)
    acc = 0
    kb, sb = bytes(k), bytes(s)
    for i in range(n):
        acc += int.from_bytes(kb[32 * i: 32 * i + 32], "big") * int.from_bytes(sb[32 * i: 32 * i + 32], "big")
    w, wf = oracle_lib.batch_mul_base(0, pyec.enc_scalar(c, acc % c.n))
    assert bytes(full) == bytes(w) and ff == int(wf[0])
    h = n // 2
    a, af = eng.lincomb(0, k[: 32 * h], pts[: 64 * h])
    b, bf = eng.lincomb(0, k[32 * h:], pts[64 * h:])
    sm, sf = eng.point_sum(0, np.concatenate([a, b]), np.array([af, bf], np.uint8))
    assert bytes(sm) == bytes(full) and sf == ff
    perm = np.random.default_rng(9).permutation(n)
    kp = k.reshape(n, 32)[perm].reshape(-1)
    pp = pts.reshape(n, 64)[perm].reshape(-1)
    p2, pf = eng.lincomb(0, kp, pp)
    assert bytes(p2) == bytes(full) and pf == ff


# ---------------------------------------------------------------------------------------------------
# batch ECDSA verification
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("curve", CURVES)
def test_ecdsa_verify_golden_vectors(eng, curve):
    """Every signature of {p256,p384,k256}/src/test_vectors/ecdsa.rs verifies; disturbing any field breaks it."""
    c = pyec.CURVES[curve]
    vec = load_golden(curve)["ecdsa"]
    z = b"".join(bytes.fromhex(v["m"]) for v in vec)
    r = b"".join(bytes.fromhex(v["r"]) for v in vec)
    s = b"".join(bytes.fromhex(v["s"]) for v in vec)
    q = b"".join(bytes.fromhex(v["q_x"]) + bytes.fromhex(v["q_y"]) for v in vec)
    assert eng.ecdsa_verify(c.cid, z, r, s, q).all()
    for field in range(4):
        bufs = [bytearray(z), bytearray(r), bytearray(s), bytearray(q)]
        width = len(bufs[field]) // len(vec)
        for i in range(len(vec)):
            bufs[field][i * width + width - 1] ^= 1 << (i % 7)
        assert not eng.ecdsa_verify(c.cid, *[bytes(b) for b in bufs]).any()
    assert eng.ecdsa_verify(c.cid, b"", b"", b"", b"").size == 0


@pytest.mark.parametrize("curve", CURVES)
def test_ecdsa_verify_vs_oracle_and_model(eng, curve):
    """Valid signatures, wrong digest / r / s / key, range failures (0, n, r + n), off-curve and out-of-range keys,
    digests >= n, both high-S policies: bit-for-bit the oracle's and the big-integer model's verdicts."""
    c = pyec.CURVES[curve]
    z, r, s, q, exp = ecdsa_pack(ecdsa_cases(c, 0x5EC1 + c.cid, nvalid=24))
    got = eng.ecdsa_verify(c.cid, z, r, s, q)
    assert bytes(got) == bytes(exp)
    assert bytes(got) == bytes(oracle_lib.ecdsa_verify(c.cid, z, r, s, q))
    got_hs = eng.ecdsa_verify(c.cid, z, r, s, q, reject_high_s=True)
    assert bytes(got_hs) == bytes(oracle_lib.ecdsa_verify(c.cid, z, r, s, q, reject_high_s=True))
    assert 0 < int(got_hs.sum()) < int(got.sum())
    # a larger mixed batch against the oracle: signatures built from k*G and (z + r d)/k with numpy-free big ints
    rng = np.random.default_rng(0xECD5A + c.cid)
    n = 600
    ds = rand_scalars(c.cid, n, 0xD0 + c.cid)
    ks = rand_scalars(c.cid, n, 0xD1 + c.cid)
    zs = rng.integers(0, 256, n * c.L, dtype=np.uint8)
    Q, _ = eng.mul_by_generator(c.cid, ds)
    R, _ = eng.mul_by_generator(c.cid, ks)
    rr, ss = bytearray(), bytearray()
    for i in range(n):
        d = int.from_bytes(bytes(ds[i * c.L:(i + 1) * c.L]), "big")
        k = int.from_bytes(bytes(ks[i * c.L:(i + 1) * c.L]), "big") or 1
        zi = int.from_bytes(bytes(zs[i * c.L:(i + 1) * c.L]), "big")
        ri = int.from_bytes(bytes(R[2 * c.L * i: 2 * c.L * i + c.L]), "big") % c.n
        si = pow(k, -1, c.n) * (zi + ri * d) % c.n
        if i % 5 == 4:
            si = (si + 1) % c.n                                  # every fifth signature is wrong
        rr += ri.to_bytes(c.L, "big"); ss += si.to_bytes(c.L, "big")
    got = eng.ecdsa_verify(c.cid, zs, bytes(rr), bytes(ss), Q)
    want = oracle_lib.ecdsa_verify(c.cid, zs, bytes(rr), bytes(ss), Q)
    assert bytes(got) == bytes(want)
    assert int(got.sum()) >= n - n // 5 - 3 and not got[4::5].any()


# ---------------------------------------------------------------------------------------------------
# BIP340 Schnorr verification and point decompression
# ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["k256_der", "k256_p1363", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_ecdsa_verify_wycheproof(eng, name):
    """The reference's Wycheproof ECDSA blobs through ecgpu_ecdsa_verify_batch (harness: k256/src/ecdsa.rs:263-384,
    ecdsa_core::new_wycheproof_test! for p256 / p384 / p224 / p521): host side = the harness's own key padding, DER /
    P1363 parsing, digest and bits2field (tests/wycheproof_lib.py); the device verdict must equal the pass flag of
    every vector that parses, and the oracle's verdict."""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    assert not [i for i, ok in p["unparsed"] if ok]
    got = eng.ecdsa_verify(c.cid, p["z"], p["r"], p["s"], p["q"], p["reject_high_s"])
    assert bytes(got) == bytes(p["expect"])
    assert bytes(got) == bytes(oracle_lib.ecdsa_verify(c.cid, p["z"], p["r"], p["s"], p["q"], p["reject_high_s"]))


def test_sm2dsa_verify_vs_reference_vector_oracle_and_model(eng):
    """ecgpu_sm2dsa_verify_batch (sm2/src/dsa/verifying.rs:138-171): the reference's own SM2DSA vector, signatures made by
    the big-int model and every way of breaking them — verdict for verdict against the expectation and the oracle; and a
    batch large enough for the pipelined host-pointer path."""
    from gpu_common import sm2dsa_cases
    e, r, s, q, exp = ecdsa_pack(sm2dsa_cases(0x5D2C))
    got = eng.sm2dsa_verify(e, r, s, q)
    assert bytes(got) == bytes(exp) == bytes(oracle_lib.sm2dsa_verify(e, r, s, q))
    reps = (1 << 19) // len(exp) + 1
    big = eng.sm2dsa_verify(e * reps, r * reps, s * reps, q * reps)
    assert bytes(big) == bytes(exp) * reps
    assert eng.sm2dsa_verify(b"", b"", b"", b"").size == 0


def test_schnorr_bip340_vectors(eng):
    """All 19 BIP340 vectors of k256/src/schnorr.rs: x-only keys lifted on the device (decompress, even y), challenge
    hashed on the host, verdicts equal to the reference's expectations and to the oracle's."""
    c = pyec.CURVES["k256"]
    vec = load_golden("k256")["schnorr"]

    def pubkey_of(sk):
        out, _ = eng.mul_by_generator(c.cid, sk)
        return bytes(out[:32])

    e, r, s, pxy, liftable, exp = schnorr_inputs(vec, lambda xs, odd: eng.decompress(c.cid, xs, odd), pubkey_of)
    assert [v["index"] for v, l in zip(vec, liftable) if not l] == [5, 14]
    pxy = pxy.reshape(-1, 64).copy()
    assert not pxy[liftable == 0].any()                               # failed lifts are zero records ...
    got = eng.schnorr_verify(e, r, s, pxy.reshape(-1))                # ... which are not on the curve: verdict 0
    assert list(got) == list(exp)
    pxy[liftable == 0] = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
    assert list(eng.schnorr_verify(e, r, s, pxy.reshape(-1)) & liftable) == list(oracle_lib.schnorr_verify(e, r, s, pxy.reshape(-1)) & liftable)
    # out-of-range signature halves and a disturbed challenge
    n = oracle_lib  # noqa: F841
    bad_s = (c.n).to_bytes(32, "big")
    one = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
    assert not eng.schnorr_verify(e[:32], r[:32], bad_s, pxy[0].tobytes()).any()
    assert not eng.schnorr_verify(e[:32], (c.p).to_bytes(32, "big"), s[:32], pxy[0].tobytes()).any()
    flipped = bytearray(e[:32]); flipped[31] ^= 1
    assert not eng.schnorr_verify(bytes(flipped), r[:32], s[:32], pxy[0].tobytes()).any()
    assert eng.schnorr_verify(e[:32], r[:32], s[:32], pxy[0].tobytes()).all()
    assert one.size == 64


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_decompress_vs_oracle(eng, curve):
    """DecompressPoint::decompress on a batch: residues and non-residues, both parities, x >= p, the generator; then
    compress(k*G) -> decompress round trip over 4096 random points."""
    c = pyec.CURVES[curve]
    rng = np.random.default_rng(0xDEC0 + c.cid)
    n = 2000
    xs = rng.integers(0, 256, n * c.L, dtype=np.uint8)
    if c.L == 66:
        xs.reshape(n, c.L)[:, 0] &= 1                                                # p521: 521-bit candidates
    xs[: c.L] = np.frombuffer((c.p).to_bytes(c.L, "big"), np.uint8)                 # x = p
    xs[c.L: 2 * c.L] = 255                                                          # x = 2^(8L) - 1
    xs[2 * c.L: 3 * c.L] = np.frombuffer(pyec.G(c)[0].to_bytes(c.L, "big"), np.uint8)
    odd = rng.integers(0, 2, n, dtype=np.uint8)
    got, gok = eng.decompress(c.cid, xs, odd)
    want, wok = oracle_lib.batch_decompress(c.cid, xs, odd)
    assert bytes(got) == bytes(want) and bytes(gok) == bytes(wok)
    expect = 0.5 * c.p / (1 << (521 if c.L == 66 else 8 * c.L))                       # candidates >= p are rejected outright
    assert gok[0] == 0 and gok[1] == 0 and gok[2] == 1 and abs(gok.mean() - expect) < 0.06
    m = 4096
    pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, m, 0xDEC1 + c.cid))
    P = pts.reshape(m, 2 * c.L)
    back, ok = eng.decompress(c.cid, P[:, : c.L].copy().reshape(-1), (P[:, 2 * c.L - 1] & 1).copy())
    assert ok.all() and bytes(back) == bytes(pts)


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_ecdh_vs_node_openssl_and_oracle(eng, curve):
    """Batch ECDH (x of k*P) against shared secrets computed by Node's crypto / OpenSSL (tests/golden/ecdh_node.json, an
    implementation independent of the reference and of this repository) and against the oracle's variable-base path."""
    import json
    import os
    from gpu_common import GOLDEN
    c = pyec.CURVES[curve]
    rows = json.load(open(os.path.join(GOLDEN, "ecdh_node.json")))[curve]
    k = b"".join(bytes.fromhex(r["d"]) for r in rows)
    p = b"".join(bytes.fromhex(r["qx"]) + bytes.fromhex(r["qy"]) for r in rows)
    x, ok = eng.ecdh(c.cid, k, p)
    assert ok.all() and bytes(x) == b"".join(bytes.fromhex(r["z"]) for r in rows)
    pub, pinf = eng.mul_by_generator(c.cid, k)                      # OpenSSL's d * G
    assert not pinf.any() and bytes(pub) == b"".join(bytes.fromhex(r["px"]) + bytes.fromhex(r["py"]) for r in rows)
    n = 500
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xECD0 + c.cid))
    ks = rand_scalars(c.cid, n, 0xECD1 + c.cid).copy()
    ks[: c.L] = 0                                                        # k = 0 -> identity -> ok = 0
    x, ok = eng.ecdh(c.cid, ks, pts)
    want, winf = oracle_lib.batch_mul(c.cid, ks, pts)
    assert bytes(x) == bytes(want.reshape(n, 2 * c.L)[:, : c.L].copy().reshape(-1))
    assert list(ok) == [0 if f else 1 for f in winf] and ok[0] == 0 and ok[1:].all()


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_uniform_schedule_entry_points(eng, curve):
    """ecgpu_batch_mul_base_ct / _mul_ct / _ecdh_ct (the reference's constant-time `mul_by_generator` / `Mul` /
    `diffie_hellman` restated one to one, ecgpu_ctmul.h) against the oracle's constant-time drivers, the golden k*G vectors,
    Node / OpenSSL's ECDH secrets and the variable-time kernels, with the variable-time entry points' error behaviour."""
    import json
    import os
    from gpu_common import GOLDEN
    ecgpu = ecgpu_module()
    c = pyec.CURVES[curve]
    G = pyec.G(c)
    # generator: golden vectors (where the reference has them), edge + ladder-corner + random scalars
    ks = edge_scalars(c) + ladder_edge_scalars(c)[:30] + comb_corner_scalars(c, 6)       # 6-bit generator windows: +-32 runs, carries
    scal = b"".join(pyec.enc_scalar(c, k) for k in ks) + bytes(rand_scalars(c.cid, 300, 0xC7EC0001 + c.cid))
    out, inf = eng.mul_by_generator(c.cid, scal, constant_time=True)
    want, winf = oracle_lib.batch_mul_base(c.cid, scal)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    if curve in CURVES:
        g = load_golden(curve)["group"]
        gk = b"".join([pyec.enc_scalar(c, v["k"]) for v in g["add"]] + [bytes.fromhex(v["k"]) for v in g["mul"]])
        out, inf = eng.mul_by_generator(c.cid, gk, constant_time=True)
        assert not inf.any() and bytes(out) == b"".join(xy(v) for v in g["add"] + g["mul"])
        gxy = pyec.enc_point(c, G)[0]
        out, inf = eng.mul(c.cid, gk, gxy * (len(gk) // c.L), constant_time=True)
        assert not inf.any() and bytes(out) == b"".join(xy(v) for v in g["add"] + g["mul"])
    # variable base: edge scalars x {G, -G, identity}, random pairs, a duplicated point
    n = 400
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xC7EC0002 + c.cid))
    pts = pts.copy()
    sc = rand_scalars(c.cid, n, 0xC7EC0003 + c.cid).copy()
    pinf = np.zeros(n, np.uint8)
    slot = 0
    for k in edge_scalars(c):
        for P in (G, pyec.neg(c, G), pyec.INF):
            e, f = pyec.enc_point(c, P)
            sc[slot * c.L: (slot + 1) * c.L] = np.frombuffer(pyec.enc_scalar(c, k), np.uint8)
            pts[slot * 2 * c.L: (slot + 1) * 2 * c.L] = np.frombuffer(e, np.uint8)
            pinf[slot] = f
            slot += 1
    assert slot < n - 2
    out, oinf = eng.mul(c.cid, sc, pts, pinf, constant_time=True)
    want, winf = oracle_lib.batch_mul(c.cid, sc, pts, pinf)                 # the oracle's constant-time `Mul`
    assert bytes(out) == bytes(want) and bytes(oinf) == bytes(winf)
    out_v, oinf_v = eng.mul(c.cid, sc, pts, pinf)                           # and the variable-time kernel
    assert bytes(out) == bytes(out_v) and bytes(oinf) == bytes(oinf_v)
    out3, _ = eng.mul(c.cid, sc[slot * c.L:], pts[slot * 2 * c.L:], None, constant_time=True)   # NULL flags
    assert bytes(out3) == bytes(want[slot * 2 * c.L:])
    # ECDH: Node / OpenSSL's shared secrets, then k = 0 -> ok = 0
    rows = json.load(open(os.path.join(GOLDEN, "ecdh_node.json")))[curve]
    k = b"".join(bytes.fromhex(r["d"]) for r in rows)
    p = b"".join(bytes.fromhex(r["qx"]) + bytes.fromhex(r["qy"]) for r in rows)
    x, ok = eng.ecdh(c.cid, k, p, constant_time=True)
    assert ok.all() and bytes(x) == b"".join(bytes.fromhex(r["z"]) for r in rows)
    ks0 = sc[slot * c.L:].copy()
    ks0[: c.L] = 0
    x, ok = eng.ecdh(c.cid, ks0, pts[slot * 2 * c.L:], constant_time=True)
    w0, wi0 = oracle_lib.batch_mul(c.cid, ks0, pts[slot * 2 * c.L:])
    assert bytes(x) == bytes(w0.reshape(-1, 2 * c.L)[:, : c.L].copy().reshape(-1)) and ok[0] == 0 and ok[1:].all()
    # errors: the verdicts of the flag pass surface as the usual codes; empty batches
    good_k = pyec.enc_scalar(c, 5)
    gxy = pyec.enc_point(c, G)[0]
    for bad_k in (c.n, 2 ** (8 * c.L) - 1):
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.mul_by_generator(c.cid, good_k + bad_k.to_bytes(c.L, "big"), constant_time=True)
        assert e.value.code == ecgpu.ERR_SCALAR_RANGE
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.mul(c.cid, good_k + bad_k.to_bytes(c.L, "big"), gxy * 2, constant_time=True)
        assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    off = bytearray(gxy); off[-1] ^= 1
    for bad_p in (bytes(off), c.p.to_bytes(c.L, "big") + gxy[c.L:]):
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.mul(c.cid, good_k * 2, gxy + bad_p, constant_time=True)
        assert e.value.code == ecgpu.ERR_POINT
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.ecdh(c.cid, good_k * 2, gxy + bad_p, constant_time=True)
        assert e.value.code == ecgpu.ERR_POINT
    # an off-curve record under a set identity flag is not an error (its verdict is dropped under the mask), as in the
    # variable-time entry point, which never looks at it
    o, f = eng.mul(c.cid, good_k * 2, gxy + bytes(off), np.array([0, 1], np.uint8), constant_time=True)
    assert f[1] == 1 and f[0] == 0
    assert eng.mul_by_generator(c.cid, b"", constant_time=True)[0].size == 0
    assert eng.mul(c.cid, b"", b"", constant_time=True)[0].size == 0


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_lincomb_constant_time_form_vs_oracle(eng, curve):
    """ecgpu_lincomb_ct = `LinearCombination::lincomb` in its constant-time form (primeorder/src/projective.rs:484-496,532-557;
    k256/src/arithmetic/mul.rs:84-98,112-163): against the oracle's `lincomb` (vartime = False: the reference's Straus loop) and
    the bucket method for n in {0, 1, 2, 3, 17, 257, 4000}, with edge scalars, identities, duplicates and cancelling pairs
    among the terms, NULL flags, the device-pointer form, and the variable-time entry point's error behaviour."""
    ecgpu = ecgpu_module()
    c = pyec.CURVES[curve]
    G = pyec.G(c)
    nmax = 4000
    pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, nmax, 0xC7EC0020 + c.cid))
    pts = pts.copy()
    sc = rand_scalars(c.cid, nmax, 0xC7EC0021 + c.cid).copy()
    pinf = np.zeros(nmax, np.uint8)
    slot = 3                                        # (the first three terms stay random: n = 1, 2, 3 are plain cases)
    for k in edge_scalars(c)[:12]:
        for P in (G, pyec.neg(c, G), pyec.INF):
            e, f = pyec.enc_point(c, P)
            sc[slot * c.L: (slot + 1) * c.L] = np.frombuffer(pyec.enc_scalar(c, k), np.uint8)
            pts[slot * 2 * c.L: (slot + 1) * 2 * c.L] = np.frombuffer(e, np.uint8)
            pinf[slot] = f
            slot += 1
    # a duplicated term and a cancelling pair (k P + (n - k) P)
    sc[(slot + 1) * c.L: (slot + 2) * c.L] = sc[slot * c.L: (slot + 1) * c.L]
    pts[(slot + 1) * 2 * c.L: (slot + 2) * 2 * c.L] = pts[slot * 2 * c.L: (slot + 1) * 2 * c.L]
    for n in (0, 1, 2, 3, 17, 257, nmax):
        s_, p_, f_ = sc[: n * c.L], pts[: n * 2 * c.L], pinf[:n]
        got, ginf = eng.lincomb_ct(c.cid, s_, p_, f_)
        want, winf = oracle_lib.msm(c.cid, s_, p_, f_, vartime=False) if n else (np.zeros(2 * c.L, np.uint8), 1)
        assert bytes(got) == bytes(want) and ginf == winf, "lincomb_ct != oracle lincomb at n = %d" % n
        vxy, vinf = eng.lincomb(c.cid, s_, p_, f_)                            # the bucket method (variable-time names)
        assert bytes(got) == bytes(vxy) and ginf == vinf
    # NULL flags on a stretch without identities
    lo = slot + 3
    got, ginf = eng.lincomb_ct(c.cid, sc[lo * c.L:(lo + 300) * c.L], pts[lo * 2 * c.L:(lo + 300) * 2 * c.L], None)
    want, winf = oracle_lib.msm(c.cid, sc[lo * c.L:(lo + 300) * c.L], pts[lo * 2 * c.L:(lo + 300) * 2 * c.L], None, vartime=False)
    assert bytes(got) == bytes(want) and ginf == winf
    # cancellation to the identity: k G + (n - k) G
    k = 0x1234567
    two = pyec.enc_scalar(c, k) + pyec.enc_scalar(c, c.n - k)
    gxy = pyec.enc_point(c, G)[0]
    got, ginf = eng.lincomb_ct(c.cid, two, gxy * 2)
    assert ginf == 1 and not got.any()
    # device-pointer form
    n = 600
    d_k, d_p, d_f = eng.to_device(sc[: n * c.L]), eng.to_device(pts[: n * 2 * c.L]), eng.to_device(pinf[:n])
    d_o, d_oi = eng.dev_alloc(2 * c.L + 64), eng.dev_alloc(16)
    eng.lincomb_ct_dev(c.cid, d_k, d_p, d_f, n, d_o, d_oi)
    want, winf = oracle_lib.msm(c.cid, sc[: n * c.L], pts[: n * 2 * c.L], pinf[:n], vartime=False)
    assert bytes(eng.to_host(d_o, 2 * c.L)) == bytes(want) and int(eng.to_host(d_oi, 1)[0]) == winf
    for b in (d_k, d_p, d_f, d_o, d_oi):
        b.free()
    # errors
    good_k = pyec.enc_scalar(c, 5)
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.lincomb_ct(c.cid, good_k + c.n.to_bytes(c.L, "little" if c.le else "big"), gxy * 2)
    assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    off = bytearray(gxy); off[-1 if not c.le else c.L] ^= 1
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.lincomb_ct(c.cid, good_k * 2, gxy + bytes(off))
    assert e.value.code == ecgpu.ERR_POINT


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_compressed_points_into_msm_and_batch_mul(eng, curve):
    """ecgpu_msm_compressed / ecgpu_batch_mul_compressed: x + SEC1 tag records are decoded on the device
    (`DecompressPoint::decompress`, primeorder/src/affine.rs:183-200, k256/src/arithmetic/affine.rs:261-280) and the ordinary
    pipeline runs on them — equal to the oracle's decompress-then-lincomb / -mul, including identity records (tag 0x00); an x that
    is on no point of the curve, x >= p and a tag outside {0, 2, 3} fail the call with ECGPU_ERR_POINT."""
    ecgpu = ecgpu_module()
    c = pyec.CURVES[curve]
    for n in (1, 2, 300, 5000):
        pts, _ = oracle_lib.batch_mul_base(c.cid, rand_scalars(c.cid, n, 0xC0EC0001 + c.cid + n))
        P = pts.reshape(n, 2 * c.L)
        ylow = P[:, c.L] if c.le else P[:, 2 * c.L - 1]                     # the byte that holds y's parity in wire order
        xs = P[:, : c.L].copy()
        tags = (2 + (ylow & 1)).astype(np.uint8)
        sc = rand_scalars(c.cid, n, 0xC0EC0002 + c.cid + n)
        pin = np.zeros(n, np.uint8)
        if n >= 300:                                                          # identity records among the terms
            for i in (5, 77, n - 1):
                xs[i] = 0; tags[i] = 0; pin[i] = 1
        got, ginf = eng.lincomb_compressed(c.cid, sc, xs.reshape(-1), tags)
        want, winf = oracle_lib.msm(c.cid, sc, pts, pin, vartime=True)
        assert bytes(got) == bytes(want) and ginf == winf, "msm_compressed != oracle at n = %d" % n
        out, oinf = eng.mul_compressed(c.cid, sc, xs.reshape(-1), tags)
        wout, woinf = oracle_lib.batch_mul(c.cid, sc, pts, pin, vartime=True)
        assert bytes(out) == bytes(wout) and bytes(oinf) == bytes(woinf)
    # the decoded records agree with the oracle's decompression (both parities of one x give P and -P)
    xs1 = xs[:64].copy().reshape(-1)
    dec, ok = oracle_lib.batch_decompress(c.cid, xs1, np.ones(64, np.uint8))
    assert ok[:5].all()
    one = np.frombuffer(pyec.enc_scalar(c, 1) * 64, np.uint8)
    out, oinf = eng.mul_compressed(c.cid, one[: 4 * c.L], xs1[: 4 * c.L], np.full(4, 3, np.uint8))
    assert bytes(out) == bytes(dec[: 4 * 2 * c.L]) and not oinf.any()
    # failures: a non-residue x, x = p, bad tags, a non-zero x under the identity tag
    rng = np.random.default_rng(0xC0EC0003 + c.cid)
    bad_x = None
    while bad_x is None:
        cand = rng.integers(0, 256, c.L, dtype=np.uint8)
        if c.L == 66:
            cand[0] &= 1
        _, okc = oracle_lib.batch_decompress(c.cid, cand, np.zeros(1, np.uint8))
        if not okc[0] and int.from_bytes(bytes(cand), "little" if c.le else "big") < c.p:
            bad_x = cand
    good_k = np.frombuffer(pyec.enc_scalar(c, 7) * 2, np.uint8)
    gx = xs[0]
    p_bytes = np.frombuffer(c.p.to_bytes(c.L, "little" if c.le else "big"), np.uint8)
    cases = [(np.concatenate([gx, bad_x]), np.array([2, 2], np.uint8)),
             (np.concatenate([gx, p_bytes]), np.array([2, 3], np.uint8)),
             (np.concatenate([gx, gx]), np.array([2, 4], np.uint8)),
             (np.concatenate([gx, gx]), np.array([2, 1], np.uint8)),
             (np.concatenate([gx, gx]), np.array([2, 0], np.uint8))]
    for x2, t2 in cases:
        for fn in (eng.lincomb_compressed, eng.mul_compressed):
            with pytest.raises(ecgpu.EcgpuError) as e:
                fn(c.cid, good_k, x2, t2)
            assert e.value.code == ecgpu.ERR_POINT
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.lincomb_compressed(c.cid, np.frombuffer((c.n).to_bytes(c.L, "little" if c.le else "big"), np.uint8), gx, np.array([2], np.uint8))
    assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    got, ginf = eng.lincomb_compressed(c.cid, b"", b"", b"")
    assert ginf == 1
    assert eng.mul_compressed(c.cid, b"", b"", b"")[0].size == 0


def test_compressed_points_device_resident_2p18_and_wipe(eng):
    """The _dev forms at 2^18 k256 / p256 terms (two-level sort, queued on an asynchronous context too) equal the x || y forms;
    ecgpu_wipe leaves the context usable."""
    for curve in ("k256", "p256"):
        c = pyec.CURVES[curve]
        n = 1 << 18
        d_s = eng.to_device(rand_scalars(c.cid, n, 0xC0EC0010 + c.cid))
        d_k = eng.to_device(rand_scalars(c.cid, n, 0xC0EC0011 + c.cid))
        d_p, d_f = eng.dev_alloc(n * 2 * c.L), eng.dev_alloc(n)
        eng.mul_by_generator_dev(c.cid, d_s, n, d_p, d_f)
        P = eng.to_host(d_p).reshape(n, 2 * c.L)
        d_x = eng.to_device(P[:, : c.L].copy().reshape(-1))
        d_t = eng.to_device((2 + (P[:, 2 * c.L - 1] & 1)).astype(np.uint8))
        d_o1, d_o2, d_i1, d_i2 = eng.dev_alloc(2 * c.L + 64), eng.dev_alloc(2 * c.L + 64), eng.dev_alloc(16), eng.dev_alloc(16)
        eng.lincomb_dev(c.cid, d_k, d_p, None, n, d_o1, d_i1)
        eng.lincomb_compressed_dev(c.cid, d_k, d_x, d_t, n, d_o2, d_i2)
        assert bytes(eng.to_host(d_o1, 2 * c.L)) == bytes(eng.to_host(d_o2, 2 * c.L)) and eng.to_host(d_i1, 1)[0] == eng.to_host(d_i2, 1)[0] == 0
        d_q1, d_q2 = eng.dev_alloc(n * 2 * c.L), eng.dev_alloc(n * 2 * c.L)
        eng.mul_dev(c.cid, d_k, d_p, None, n, d_q1, d_f)
        eng.set_async(True)
        eng.mul_compressed_dev(c.cid, d_k, d_x, d_t, n, d_q2, d_f)
        eng.synchronize()
        eng.set_async(False)
        assert bytes(eng.to_host(d_q1)) == bytes(eng.to_host(d_q2))
        eng.wipe()
        eng.lincomb_compressed_dev(c.cid, d_k, d_x, d_t, n, d_o2, d_i2)
        assert bytes(eng.to_host(d_o1, 2 * c.L)) == bytes(eng.to_host(d_o2, 2 * c.L))
        for b in (d_s, d_k, d_p, d_f, d_x, d_t, d_o1, d_o2, d_i1, d_i2, d_q1, d_q2):
            b.free()


def test_uniform_schedule_device_resident_and_queued(eng):
    """The _dev forms on device-resident p256 / k256 batches (2^14 elements: several waves per SIMD slot, the table scratch
    reused by the lanes' later elements), synchronous and queued (ecgpu_set_async), equal to the variable-time results."""
    for curve in ("k256", "p256"):
        c = pyec.CURVES[curve]
        n = 1 << 14
        ks = rand_scalars(c.cid, n, 0xC7EC0010 + c.cid)
        d_k = eng.to_device(ks)
        d_pts = eng.dev_alloc(n * 2 * c.L)
        d_out = eng.dev_alloc(n * 2 * c.L)
        d_out_ct = eng.dev_alloc(n * 2 * c.L)
        d_inf = eng.dev_alloc(n)
        eng.mul_by_generator_dev(c.cid, d_k, n, d_pts, d_inf)
        eng.mul_by_generator_dev(c.cid, d_k, n, d_out_ct, d_inf, constant_time=True)
        assert bytes(eng.to_host(d_pts)) == bytes(eng.to_host(d_out_ct))
        d_k2 = eng.to_device(rand_scalars(c.cid, n, 0xC7EC0011 + c.cid))
        eng.mul_dev(c.cid, d_k2, d_pts, None, n, d_out, d_inf)
        eng.set_async(True)
        eng.mul_dev(c.cid, d_k2, d_pts, None, n, d_out_ct, d_inf, constant_time=True)
        eng.synchronize()
        eng.set_async(False)
        assert bytes(eng.to_host(d_out)) == bytes(eng.to_host(d_out_ct))
        for b in (d_k, d_k2, d_pts, d_out, d_out_ct, d_inf):
            b.free()


def test_msm_lanes_several_in_flight(eng):
    """ecgpu_set_msm_lanes(2 .. 4) on an asynchronous context: consecutive MSMs rotate over that many internal streams and
    workspaces.  Eight queued MSMs of different sizes (both sorts, GLV and plain, two curves) on points that an earlier QUEUED
    generator multiplication produces (the lane must start after it) equal their synchronous results; a bad scalar in one of
    them surfaces at synchronize(); lanes = 1 restores the single stream; other values are refused."""
    ecgpu = ecgpu_module()
    jobs = []
    for curve, n in (("k256", 5000), ("p256", 3000), ("k256", 1 << 17), ("k256", 77), ("p256", 1 << 16), ("k256", 20000),
                     ("p256", 1), ("k256", 1 << 18)):
        c = pyec.CURVES[curve]
        d_s = eng.to_device(rand_scalars(c.cid, n, 0x1A9E0 + n))
        d_k = eng.to_device(rand_scalars(c.cid, n, 0x1A9E1 + n))
        d_p, d_f = eng.dev_alloc(n * 2 * c.L), eng.dev_alloc(n)
        outs = [(eng.dev_alloc(2 * c.L), eng.dev_alloc(16)) for _ in range(2)]
        jobs.append((c, n, d_s, d_k, d_p, d_f, outs))
    want = []
    for c, n, d_s, d_k, d_p, d_f, outs in jobs:                          # synchronous reference
        eng.mul_by_generator_dev(c.cid, d_s, n, d_p, d_f)
        eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[0])
        want.append((bytes(eng.to_host(outs[0][0], 2 * c.L)), int(eng.to_host(outs[0][1], 1)[0])))
    eng.set_async(True)
    for nl in (2, 3, 4):
        eng.set_msm_lanes(nl)
        for c, n, d_s, d_k, d_p, d_f, outs in jobs:                      # points re-made by queued work, then the MSM on a lane
            eng.to_device(np.zeros(16, np.uint8), outs[1][0])
            eng.mul_by_generator_dev(c.cid, d_s, n, d_p, d_f)
            eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[1])
        eng.synchronize()
        for (c, n, d_s, d_k, d_p, d_f, outs), w in zip(jobs, want):
            assert (bytes(eng.to_host(outs[1][0], 2 * c.L)), int(eng.to_host(outs[1][1], 1)[0])) == w, (nl, c.name, n)
    eng.set_msm_lanes(2)
    # a scalar >= n in a queued lane MSM: the error is deferred to synchronize()
    c, n, d_s, d_k, d_p, d_f, outs = jobs[0]
    bad = np.frombuffer(bytes(rand_scalars(c.cid, n, 5)), np.uint8).copy()
    bad[: c.L] = 0xFF
    d_bad = eng.to_device(bad)
    eng.lincomb_dev(c.cid, d_bad, d_p, None, n, *outs[1])
    eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[0])
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.synchronize()
    assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    assert bytes(eng.to_host(outs[0][0], 2 * c.L)) == want[0][0]
    # ordering rule of include/ecgpu.h: any OTHER entry point called after a lane MSM waits for the lanes first — a copy to the
    # host straight after the queued MSMs (no ecgpu_synchronize in between) sees their results, and a later overwrite of the
    # inputs by a queued batch call cannot overtake the MSM that still reads them
    eng.set_msm_lanes(3)
    big = [2, 4, 7]                                                       # the jobs that run the bucket method, i.e. on a lane
    for j in big:
        eng.to_device(np.zeros(16, np.uint8), jobs[j][6][1][0])
    for j in big:
        c, n, d_s, d_k, d_p, d_f, outs = jobs[j]
        eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[1])
    for j in big:
        c, n, d_s, d_k, d_p, d_f, outs = jobs[j]
        assert bytes(eng.to_host(outs[1][0], 2 * c.L)) == want[j][0], ("copy after lane MSM", c.name, n)
    c, n, d_s, d_k, d_p, d_f, outs = jobs[2]
    eng.to_device(np.zeros(16, np.uint8), outs[1][0])
    eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[1])
    eng.mul_by_generator_dev(c.cid, d_k, n, d_p, d_f)                    # overwrites the MSM's points: must wait for the lane
    eng.synchronize()
    assert bytes(eng.to_host(outs[1][0], 2 * c.L)) == want[2][0]
    eng.mul_by_generator_dev(c.cid, d_s, n, d_p, d_f)                    # (the points as they were)
    eng.synchronize()
    c, n, d_s, d_k, d_p, d_f, outs = jobs[0]
    with pytest.raises(ecgpu.EcgpuError):
        eng.set_msm_lanes(5)
    eng.set_msm_lanes(1)
    eng.lincomb_dev(c.cid, d_k, d_p, None, n, *outs[1])
    eng.synchronize()
    eng.set_async(False)
    assert bytes(eng.to_host(outs[1][0], 2 * c.L)) == want[0][0]


def test_schnorr_bip340_vectors_from_wire_bytes(eng):
    """`VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)` entirely on the device (lift_x, SHA-256 tagged hash, s G - e P):
    the 19 BIP340 vectors of k256/src/schnorr.rs, the 32-byte-message ones as one batch, plus random batches against the
    oracle for message lengths on both sides of the SHA-256 block boundaries."""
    c = pyec.CURVES["k256"]
    vec = load_golden("k256")["schnorr"]
    pk_of = lambda v: bytes.fromhex(v["public_key"]) if "public_key" in v else bytes(eng.mul_by_generator(c.cid, bytes.fromhex(v["secret_key"]))[0][:32])
    v32 = [v for v in vec if len(v["message"]) == 64]
    got = eng.schnorr_verify_raw(b"".join(pk_of(v) for v in v32), b"".join(bytes.fromhex(v["message"]) for v in v32), 32,
                                 b"".join(bytes.fromhex(v["signature"]) for v in v32))
    assert list(got) == [1 if v["valid"] else 0 for v in v32] and len(v32) == 15
    for v in vec:
        msg = bytes.fromhex(v["message"])
        assert int(eng.schnorr_verify_raw(pk_of(v), msg, len(msg), bytes.fromhex(v["signature"]))[0]) == (1 if v["valid"] else 0), v["index"]
    rng = np.random.default_rng(0xB1F)
    valid_sig = bytes.fromhex(vec[1]["signature"]); valid_pk = pk_of(vec[1])
    for mlen in (0, 31, 55, 56, 64, 119, 120, 300):
        n = 64
        pk = rng.integers(0, 256, n * 32, dtype=np.uint8); msgs = rng.integers(0, 256, max(1, n * mlen), dtype=np.uint8)
        sigs = rng.integers(0, 256, n * 64, dtype=np.uint8)
        pk[:32] = np.frombuffer(valid_pk, np.uint8); sigs[:64] = np.frombuffer(valid_sig, np.uint8)
        got = eng.schnorr_verify_raw(pk, msgs, mlen, sigs)
        assert bytes(got) == bytes(oracle_lib.schnorr_verify_raw(pk, msgs, mlen, sigs)), mlen


def test_pinned_host_buffers(eng):
    """ecgpu_host_alloc / ecgpu_host_free: the host-pointer entry points on page-locked buffers give the same bytes."""
    c = pyec.CURVES["k256"]
    n = 5000
    scal = rand_scalars(c.cid, n, 0xEC0000F1)
    want, winf = eng.mul_by_generator(c.cid, scal)
    pin_s, pin_o, pin_i = eng.host_alloc(n * c.L), eng.host_alloc(n * 2 * c.L), eng.host_alloc(n)
    pin_s[:] = scal
    out, inf = eng.mul_by_generator(c.cid, pin_s, out=pin_o, inf=pin_i)
    assert out.ctypes.data == pin_o.ctypes.data
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    for a in (pin_s, pin_o, pin_i):
        eng.host_free(a)


@pytest.mark.parametrize("curve", ["k256", "p384"])
def test_pipelined_host_pointer_calls(eng, curve):
    """From 2^19 units on, the host-pointer batch calls run as a three-stage pipeline over chunks of 2^18 units (upload
    thread / device-pointer entry point / download thread): ragged last chunk, identity points, NULL flags, and the
    scalar-range error of a late chunk must come through."""
    c = pyec.CURVES[curve]
    n = (1 << 19) + 4321
    scal = rand_scalars(c.cid, n, 0xEC0000C1 + c.cid).copy()
    scal[: c.L] = 0                                                        # k = 0 -> identity in chunk 0
    scal[(n - 1) * c.L: n * c.L] = np.frombuffer(pyec.enc_scalar(c, c.n - 1), np.uint8)
    out, inf = eng.mul_by_generator(c.cid, scal)
    assert inf[0] == 1 and inf[1:].sum() == 0
    idx = np.concatenate([np.arange(0, 300), np.arange((1 << 18) - 150, (1 << 18) + 150), np.arange(n - 300, n)])
    pick = lambda a, w: np.ascontiguousarray(a.reshape(n, w)[idx]).reshape(-1)
    want, winf = oracle_lib.batch_mul_base(c.cid, pick(scal, c.L))
    assert bytes(pick(out, 2 * c.L)) == bytes(want) and bytes(inf[idx]) == bytes(winf)
    # variable base on the points just computed, a few of them replaced by the identity
    pinf = np.zeros(n, np.uint8)
    pinf[[0, 5, (1 << 18) + 7, n - 2]] = 1
    pts = out.copy()
    pts.reshape(n, 2 * c.L)[pinf == 1] = 0
    k2 = rand_scalars(c.cid, n, 0xEC0000D1 + c.cid)
    o2, i2 = eng.mul(c.cid, k2, pts, pinf)
    w2, wi2 = oracle_lib.batch_mul(c.cid, pick(k2, c.L), pick(pts, 2 * c.L), pinf[idx])
    assert bytes(pick(o2, 2 * c.L)) == bytes(w2) and bytes(i2[idx]) == bytes(wi2)
    assert i2[5] == 1 and i2[(1 << 18) + 7] == 1
    ecgpu = ecgpu_module()
    bad = scal.copy()
    bad[(n - 5) * c.L: (n - 4) * c.L] = 0xFF                                # >= n, in the last chunk
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.mul_by_generator(c.cid, bad)
    assert e.value.code == ecgpu.ERR_SCALAR_RANGE


@pytest.mark.parametrize("name", ["k256_der", "k256_p1363", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_ecdsa_verify_messages_wycheproof(eng, name):
    """The reference's Wycheproof blobs at the message level through ecgpu_ecdsa_verify_msg_batch — `Verifier::verify(msg,
    &sig)`: the curve's digest (SHA-256 / 384 / 224 / 512) and bits2field on the device, one call per message length; every
    verdict equals the pass flag and the oracle's."""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    for ln, (idx, q, m, sg) in wycheproof_lib.by_message_length(p).items():
        got = eng.ecdsa_verify_msg(c.cid, q, m, ln, sg, p["reject_high_s"])
        assert bytes(got) == bytes(p["expect"][idx]), (name, ln)
        assert bytes(got) == bytes(oracle_lib.ecdsa_verify_msg(c.cid, q, m, ln, sg, p["reject_high_s"]))


@pytest.mark.parametrize("curve", [c for c in ALL_CURVES if c not in ("sm2", "p192")])
def test_ecdsa_verify_messages_vs_oracle(eng, curve):
    """Model-made signatures over messages whose lengths put the padding on every side of a block boundary (64-byte blocks for
    SHA-224 / 256, 128-byte blocks for SHA-384 / 512), every third one broken: the oracle's verdicts; a 2^19 + 333 batch through
    the pipelined host path; no digest for p192 / sm2 / bign256."""
    import hashlib
    import random
    import wycheproof_lib
    ecgpu = ecgpu_module()
    H = {"k256": "sha256", "p256": "sha256", "p384": "sha384", "p224": "sha224", "p521": "sha512", "bp256": "sha256", "bp384": "sha384",
         "bp256t1": "sha256", "bp384t1": "sha384"}
    c = pyec.CURVES[curve]
    L = c.L
    rng = random.Random(0x3E55 + c.cid)
    G = pyec.G(c)
    for msg_len in (0, 1, 55, 56, 64, 111, 112, 119, 120, 128, 200):
        q = m = sg = b""
        exp = []
        for i in range(6):
            d, k = rng.randrange(1, c.n), rng.randrange(1, c.n)
            msg = bytes(rng.randrange(256) for _ in range(msg_len))
            z = int.from_bytes(wycheproof_lib.bits2field(hashlib.new(H[curve], msg).digest(), L), "big")
            r, s_ = pyec.ecdsa_sign(c, d, z, k)
            if i % 3 == 2:
                s_ = (s_ + 1) % c.n or 1
            q += pyec.enc_point(c, pyec.mul(c, d, G))[0]; m += msg; sg += r.to_bytes(L, "big") + s_.to_bytes(L, "big")
            exp.append(0 if i % 3 == 2 else 1)
        got = eng.ecdsa_verify_msg(c.cid, q, m, msg_len, sg)
        assert list(got) == exp, (curve, msg_len)
        assert bytes(got) == bytes(oracle_lib.ecdsa_verify_msg(c.cid, q, m, msg_len, sg))
    if curve in ("k256", "p521"):
        reps = ((1 << 19) + 333) // 6 + 1
        big = eng.ecdsa_verify_msg(c.cid, q * reps, m * reps, msg_len, sg * reps)
        assert bytes(big) == bytes(np.tile(np.array(exp, np.uint8), reps))
    assert eng.ecdsa_verify_msg(c.cid, b"", b"", 7, b"").size == 0
    if curve == "k256":
        for name in ("p192", "sm2", "bign256"):
            cid = pyec.CURVES[name].cid
            l2 = ecgpu.FIELD_BYTES[cid]
            with pytest.raises(ecgpu.EcgpuError) as e:
                eng.ecdsa_verify_msg(cid, bytes(2 * l2), b"abc", 3, bytes(2 * l2))
            assert e.value.code == ecgpu.ERR_CURVE


def test_sm2dsa_verify_messages_vs_reference_vector_oracle_and_model(eng):
    """ecgpu_sm2dsa_verify_msg_batch — `VerifyingKey::new(distid, pk)?.verify(msg, sig)` with Z and e = SM3(Z || M) hashed on
    the device: the reference's message-level vector (sm2/tests/sm2dsa.rs:16-35) verifies, another identifier / message does
    not; model-made signatures and broken ones under four identifiers (empty ... 200 bytes) and message lengths that put the
    SM3 padding on every side of a block boundary; a 70,000-element batch; the empty batch; an over-long identifier."""
    from gpu_common import SM2DSA_KAT as K, sm2dsa_msg_cases, sm2dsa_msg_pack
    ecgpu = ecgpu_module()
    pk, sig, msg = bytes.fromhex(K["public_key"])[1:], bytes.fromhex(K["signature"]), K["message"]
    assert eng.sm2dsa_verify_msg(K["identity"], pk, msg, len(msg), sig)[0] == 1
    assert eng.sm2dsa_verify_msg(K["identity"] + b"x", pk, msg, len(msg), sig)[0] == 0
    assert eng.sm2dsa_verify_msg(K["identity"], pk, b"testinh", len(msg), sig)[0] == 0
    for distid, msg_len in ((b"", 0), (b"1234567812345678", 32), (bytes(range(100)), 77), (bytes(200), 23), (b"id", 150)):
        q, m, sg, exp = sm2dsa_msg_pack(sm2dsa_msg_cases(0x5D50 + msg_len, distid, msg_len))
        got = eng.sm2dsa_verify_msg(distid, q, m, msg_len, sg)
        assert bytes(got) == bytes(exp) == bytes(oracle_lib.sm2dsa_verify_msg(distid, q, m, msg_len, sg))
    reps = 70000 // len(exp) + 1
    big = eng.sm2dsa_verify_msg(distid, q * reps, m * reps, msg_len, sg * reps)
    assert bytes(big) == bytes(np.tile(exp, reps))
    reps = ((1 << 19) + 777) // len(exp) + 1                     # above the pipeline threshold of the host-pointer entry
    big = eng.sm2dsa_verify_msg(distid, q * reps, m * reps, msg_len, sg * reps)
    assert bytes(big) == bytes(np.tile(exp, reps))
    assert eng.sm2dsa_verify_msg(b"x", b"", b"", 5, b"").size == 0
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.sm2dsa_verify_msg(bytes(8192), pk, msg, len(msg), sig)
    assert e.value.code == ecgpu.ERR_ARG


def test_bign_verify_vs_reference_vector_oracle_and_model(eng):
    """ecgpu_bign_verify_batch / ecgpu_bign_verify_msg_batch (bignp256/src/ecdsa/verifying.rs:100-169, belt-hash on the device):
    the reference's own signature vector (bignp256/tests/ecdsa.rs:21-46) verifies at both levels and a changed message does not;
    model-made signatures (hashes on both sides of q, S1 + H on both sides of q) and every way of breaking one — S0 = 0, S1 = 0,
    S1 >= q, key off the curve / out of range / all zero, R = O — verdict for verdict against the expectation and the oracle;
    message lengths on every side of the 32-byte block boundary; batches above the pipeline threshold of the host-pointer
    entries; device-resident buffers; the empty batch."""
    from gpu_common import BIGN_KAT as K, bign_cases, bign_msg_cases
    pk, sig, msg = bytes.fromhex(K["public_key"]), bytes.fromhex(K["signature"]), bytes.fromhex(K["message"])
    assert eng.bign_verify_msg(pk, msg, len(msg), sig)[0] == 1
    assert eng.bign_verify_msg(pk, msg[:-1] + b"\x59", len(msg), sig)[0] == 0
    assert eng.bign_verify(pyec.belt_hash(msg), sig, pk)[0] == 1
    cases = bign_cases(0xB16C)
    h, sg, q = (b"".join(c[k] for c in cases) for k in range(3))
    exp = bytes(int(c[3]) for c in cases)
    assert sum(exp) >= 9 and exp.count(0) > 40
    got = eng.bign_verify(h, sg, q)
    assert bytes(got) == exp == bytes(oracle_lib.bign_verify(h, sg, q))
    reps = ((1 << 19) + 333) // len(exp) + 1
    assert bytes(eng.bign_verify(h * reps, sg * reps, q * reps)) == exp * reps
    n = len(exp)
    d_h, d_s, d_q, d_ok = eng.to_device(h), eng.to_device(sg), eng.to_device(q), eng.dev_alloc(n)
    eng.bign_verify_dev(d_h, d_s, d_q, n, d_ok)
    assert bytes(eng.to_host(d_ok, n)) == exp
    for msg_len in (0, 1, 13, 31, 32, 33, 64, 77, 150):
        mc = bign_msg_cases(0xB170 + msg_len, msg_len)
        q, m, sg = (b"".join(c[k] for c in mc) for k in range(3))
        exp = bytes(int(c[3]) for c in mc)
        got = eng.bign_verify_msg(q, m, msg_len, sg)
        assert bytes(got) == exp == bytes(oracle_lib.bign_verify_msg(q, m, msg_len, sg)), msg_len
    reps = ((1 << 19) + 777) // len(exp) + 1
    assert bytes(eng.bign_verify_msg(q * reps, m * reps, msg_len, sg * reps)) == exp * reps
    n = len(exp)
    d_q, d_m, d_s, d_ok2 = eng.to_device(q), eng.to_device(m), eng.to_device(sg), eng.dev_alloc(n)
    eng.bign_verify_msg_dev(d_q, d_m, msg_len, d_s, n, d_ok2)
    assert bytes(eng.to_host(d_ok2, n)) == exp
    assert eng.bign_verify(b"", b"", b"").size == 0 and eng.bign_verify_msg(b"", b"", 5, b"").size == 0


# ---- ECDSA public-key recovery (ecgpu_ecdsa_recover_batch) ------------------------------------------------------------------
def test_ecdsa_recover_reference_vectors(eng):
    """The reference's recovery vectors (k256/src/ecdsa.rs:190-211 RECOVERY_TEST_VECTORS, :233-261 the Ethereum example)
    through the C ABI: the stated keys come out; the other parity recovers other keys; error codes for the curves that have
    no ECDSA recovery; the empty batch."""
    from gpu_common import recovery_golden, recover_pack
    ecgpu = ecgpu_module()
    z, r, s, recid, exp_xy, _ = recover_pack(recovery_golden(), 32)
    out, ok = eng.ecdsa_recover(ecgpu.K256, z, r, s, recid, reject_high_s=True)
    assert ok.all() and bytes(out) == exp_xy
    want, wok = oracle_lib.ecdsa_recover(0, z, r, s, recid ^ 1, True)
    out2, ok2 = eng.ecdsa_recover(ecgpu.K256, z, r, s, recid ^ 1, reject_high_s=True)
    assert bytes(ok2) == bytes(wok) and bytes(out2) == bytes(want) and bytes(out2) != exp_xy
    assert eng.ecdsa_verify(ecgpu.K256, z, r, s, out, reject_high_s=True).all()     # what recover_from_prehash ends with
    out0, ok0 = eng.ecdsa_recover(ecgpu.K256, b"", b"", b"", b"")
    assert out0.size == 0 and ok0.size == 0
    for cid in (pyec.CURVES["sm2"].cid, pyec.CURVES["bign256"].cid):
        L = ecgpu.FIELD_BYTES[cid]
        with pytest.raises(ecgpu.EcgpuError) as e:
            eng.ecdsa_recover(cid, bytes(L), bytes(L), bytes(L), b"\0")
        assert e.value.code == ecgpu.ERR_CURVE


@pytest.mark.parametrize("curve", [c for c in ALL_CURVES if c != "sm2"])
def test_ecdsa_recover_vs_oracle_and_model(eng, curve):
    """Recovery ids of the nonce point and the other three, disturbed fields, range failures, ids above 3, candidates off
    the curve, x-reduced candidates, both high-S policies: key for key the oracle's restatement of `recover_from_prehash`
    and the big-integer model's; then 600 signatures made with the engine itself (every fifth with the wrong parity)."""
    from gpu_common import recover_cases, recover_pack
    c = pyec.CURVES[curve]
    L = c.L
    cases = recover_cases(c, 0x4EC2 + c.cid, nvalid=12)
    z, r, s, recid, exp_xy, exp_ok = recover_pack(cases, L)
    out, ok = eng.ecdsa_recover(c.cid, z, r, s, recid)
    assert bytes(ok) == bytes(exp_ok) and bytes(out) == exp_xy
    for high in (False, True):
        want, wok = oracle_lib.ecdsa_recover(c.cid, z, r, s, recid, high)
        got, gok = eng.ecdsa_recover(c.cid, z, r, s, recid, reject_high_s=high)
        assert bytes(gok) == bytes(wok) and bytes(got) == bytes(want)
    assert 0 < int(gok.sum()) < int(ok.sum()) < len(cases)
    n = 600
    ds = rand_scalars(c.cid, n, 0xD4 + c.cid)
    ks = rand_scalars(c.cid, n, 0xD5 + c.cid)
    zs = np.random.default_rng(0xEC4EC + c.cid).integers(0, 256, n * L, dtype=np.uint8)
    if c.n.bit_length() < 8 * L:
        zs.reshape(n, L)[:, 0] = 0                                  # p521: digests below 2^520
    Q, _ = eng.mul_by_generator(c.cid, ds)
    R, _ = eng.mul_by_generator(c.cid, ks)
    rr, ss, ids = bytearray(), bytearray(), bytearray()
    for i in range(n):
        d = int.from_bytes(bytes(ds[i * L:(i + 1) * L]), "big")
        k = int.from_bytes(bytes(ks[i * L:(i + 1) * L]), "big") or 1
        zi = int.from_bytes(bytes(zs[i * L:(i + 1) * L]), "big")
        x = int.from_bytes(bytes(R[2 * L * i: 2 * L * i + L]), "big")
        ri = x % c.n
        si = pow(k, -1, c.n) * (zi + ri * d) % c.n
        rr += ri.to_bytes(L, "big"); ss += si.to_bytes(L, "big")
        ids.append((int(R[2 * L * i + 2 * L - 1] & 1) ^ (1 if i % 5 == 4 else 0)) | (2 if x >= c.n else 0))
    got, gok = eng.ecdsa_recover(c.cid, zs, bytes(rr), bytes(ss), bytes(ids))
    want, wok = oracle_lib.ecdsa_recover(c.cid, zs, bytes(rr), bytes(ss), bytes(ids))
    assert bytes(gok) == bytes(wok) and bytes(got) == bytes(want)
    good = np.ones(n, bool); good[4::5] = False
    Qr, gr = np.asarray(Q).reshape(n, 2 * L), np.asarray(got).reshape(n, 2 * L)
    nonzero = np.array([any(ss[i * L:(i + 1) * L]) and any(rr[i * L:(i + 1) * L]) for i in range(n)])
    assert (gr[good & nonzero] == Qr[good & nonzero]).all() and not (gr[~good] == Qr[~good]).all(axis=1).any()


@pytest.mark.parametrize("name", ["k256_der", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_ecdsa_recover_wycheproof(eng, name):
    """The reference's Wycheproof blobs through ecgpu_ecdsa_recover_batch: every parsed vector under all four recovery ids; one
    of them gives back the vector's public key exactly when the vector is valid; keys and verdicts equal the oracle's."""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    z, r, s, ids = wycheproof_lib.recovery_batch(p)
    keys, ok = eng.ecdsa_recover(c.cid, z, r, s, ids, reject_high_s=p["reject_high_s"])
    assert bytes(wycheproof_lib.recovery_matches(p, keys, ok)) == bytes(p["expect"])
    wk, wok = oracle_lib.ecdsa_recover(c.cid, z, r, s, ids, p["reject_high_s"])
    assert bytes(keys) == bytes(wk) and bytes(ok) == bytes(wok)


def test_ecdsa_recover_device_resident_2p20(eng):
    """2^20 k256 signatures, device-resident through ecgpu_ecdsa_recover_batch_dev: the corner-case set tiled over the batch
    (every chunk of the kernels sees every case) equals the oracle's keys and verdicts, tiled the same way."""
    from gpu_common import recover_cases, recover_pack
    c = pyec.CURVES["k256"]
    L, n = c.L, 1 << 20
    cases = recover_cases(c, 0x4EC3, nvalid=12)
    z, r, s, recid, exp_xy, exp_ok = recover_pack(cases, L)
    m = len(cases)
    def tile(b, unit):
        a = np.frombuffer(bytes(b), np.uint8).reshape(-1, unit)
        return np.ascontiguousarray(np.tile(a, ((n + m - 1) // m, 1))[:n]).reshape(-1)
    Z, R, S, I = tile(z, L), tile(r, L), tile(s, L), tile(recid, 1)
    d_z, d_r, d_s, d_i = eng.to_device(Z), eng.to_device(R), eng.to_device(S), eng.to_device(I)
    d_o, d_ok = eng.dev_alloc(n * 2 * L), eng.dev_alloc(n + 16)
    eng.ecdsa_recover_dev(c.cid, d_z, d_r, d_s, d_i, n, True, d_o, d_ok)
    out, ok = eng.to_host(d_o), eng.to_host(d_ok, n)
    want, wok = oracle_lib.ecdsa_recover(c.cid, z, r, s, recid, True)
    assert bytes(ok) == bytes(tile(wok, 1)) and bytes(out) == bytes(tile(want, 2 * L))
    assert 0 < int(wok.sum()) < m
    for b in (d_z, d_r, d_s, d_i, d_o, d_ok):
        b.free()


def test_pipelined_host_pointer_calls_other_entry_points(eng):
    """ECDSA / Schnorr verification, ECDH, decompression and a*G + b*P above the pipeline threshold: the big call must
    return exactly what two calls on the halves (both below the threshold, i.e. the serial path) return, and the known
    verdicts of the reference's vectors, tiled across all chunks, must come out."""
    c = pyec.CURVES["k256"]
    L = c.L
    n = (1 << 19) + 1500
    h = n // 2
    cat = lambda a, b: np.concatenate([np.asarray(a), np.asarray(b)])
    def tile(b, unit):
        a = np.frombuffer(b, np.uint8).reshape(-1, unit)
        return np.ascontiguousarray(np.tile(a, ((n + a.shape[0] - 1) // a.shape[0], 1))[:n]).reshape(-1)
    # ECDSA: the corner-case set (valid and broken signatures), tiled
    z, r, s_, q, exp = ecdsa_pack(ecdsa_cases(c, 0xE1))
    Z, R, S, Q = tile(z, L), tile(r, L), tile(s_, L), tile(q, 2 * L)
    got = eng.ecdsa_verify(c.cid, Z, R, S, Q)
    assert bytes(got) == bytes(np.tile(exp, (n + len(exp) - 1) // len(exp))[:n])
    assert bytes(got) == bytes(cat(eng.ecdsa_verify(c.cid, Z[: h * L], R[: h * L], S[: h * L], Q[: h * 2 * L]),
                                   eng.ecdsa_verify(c.cid, Z[h * L:], R[h * L:], S[h * L:], Q[h * 2 * L:])))
    # public-key recovery: the corner-case set, tiled (keys and verdicts)
    from gpu_common import recover_cases, recover_pack
    rz, rr_, rs, rid, rxy, rok = recover_pack(recover_cases(c, 0xE2), L)
    RZ, RR, RS, RI = tile(rz, L), tile(rr_, L), tile(rs, L), tile(bytes(rid), 1)
    gxy, gok = eng.ecdsa_recover(c.cid, RZ, RR, RS, RI)
    assert bytes(gok) == bytes(tile(bytes(rok), 1)) and bytes(gxy) == bytes(tile(rxy, 2 * L))
    # BIP340 from wire bytes: the 15 vectors with 32-byte messages, tiled
    vec = [v for v in load_golden("k256")["schnorr"] if len(v["message"]) == 64]
    pk_of = lambda v: bytes.fromhex(v["public_key"]) if "public_key" in v else bytes(eng.mul_by_generator(c.cid, bytes.fromhex(v["secret_key"]))[0][:32])
    PK = tile(b"".join(pk_of(v) for v in vec), 32)
    MS = tile(b"".join(bytes.fromhex(v["message"]) for v in vec), 32)
    SG = tile(b"".join(bytes.fromhex(v["signature"]) for v in vec), 64)
    got = eng.schnorr_verify_raw(PK, MS, 32, SG)
    want = np.array([1 if v["valid"] else 0 for v in vec], np.uint8)
    assert bytes(got) == bytes(np.tile(want, (n + len(vec) - 1) // len(vec))[:n])
    # ECDH, decompression, a*G + b*P: big call == two serial calls
    k = rand_scalars(c.cid, n, 0xEC0000E1)
    pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xEC0000E2))
    x, ok = eng.ecdh(c.cid, k, pts)
    x1, ok1 = eng.ecdh(c.cid, k[: h * L], pts[: h * 2 * L]); x2, ok2 = eng.ecdh(c.cid, k[h * L:], pts[h * 2 * L:])
    assert bytes(x) == bytes(cat(x1, x2)) and bytes(ok) == bytes(cat(ok1, ok2)) and ok.all()
    xs = np.ascontiguousarray(pts.reshape(n, 2 * L)[:, :L]).reshape(-1).copy()
    xs[5 * L: 6 * L] = 0xFF                                                    # not a field element
    odd = (pts.reshape(n, 2 * L)[:, 2 * L - 1] & 1).astype(np.uint8)
    d, dok = eng.decompress(c.cid, xs, odd)
    assert dok[5] == 0 and dok.sum() == n - 1
    keep = np.ones(n, bool); keep[5] = False
    assert bytes(d.reshape(n, 2 * L)[keep]) == bytes(pts.reshape(n, 2 * L)[keep])
    a = rand_scalars(c.cid, n, 0xEC0000E3)
    o, oi = eng.mul_by_generator_and_mul_add(c.cid, a, k, pts)
    o1, oi1 = eng.mul_by_generator_and_mul_add(c.cid, a[: h * L], k[: h * L], pts[: h * 2 * L])
    o2, oi2 = eng.mul_by_generator_and_mul_add(c.cid, a[h * L:], k[h * L:], pts[h * 2 * L:])
    assert bytes(o) == bytes(cat(o1, o2)) and bytes(oi) == bytes(cat(oi1, oi2))


def test_host_pointer_msm_in_chunks(eng):
    """ecgpu_msm on host buffers of 2^23 terms and more: one MSM per chunk of 2^22 terms under the upload of the next
    chunk, partial sums added at the end.  All points = G: the result must be (sum k_i) G; and with a few terms switched
    to the identity, the sum without them."""
    c = pyec.K256
    n = (1 << 23) + 999
    k = rand_scalars(c.cid, n, 0xEC0000F7)
    gxy = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)
    pts = np.tile(gxy, n)
    o, f = eng.lincomb(c.cid, k, pts)
    w, wf = oracle_lib.batch_mul_base(c.cid, pyec.enc_scalar(c, scalars_to_int_sum(k, c.L, c.n)))
    assert bytes(o) == bytes(w) and f == int(wf[0])
    inf = np.zeros(n, np.uint8)
    drop = [0, 1, (1 << 22) - 1, 1 << 22, (1 << 23) + 500, n - 1]
    inf[drop] = 1
    o2, f2 = eng.lincomb(c.cid, k, pts, inf)
    rest = (scalars_to_int_sum(k, c.L, c.n) - sum(int.from_bytes(bytes(k[i * c.L: (i + 1) * c.L]), "big") for i in drop)) % c.n
    w2, wf2 = oracle_lib.batch_mul_base(c.cid, pyec.enc_scalar(c, rest))
    assert bytes(o2) == bytes(w2) and f2 == int(wf2[0])


@pytest.mark.parametrize("curve", ["k256", "p256", "p384", "p224", "bp256"])
@pytest.mark.parametrize("n,shards", [(5000, 2), ((1 << 17) + 77, 3), (3, 4)])
def test_msm_parts_and_finish_split(eng, curve, n, shards):
    """The two halves of a multi-GPU MSM on one GPU: ecgpu_msm_parts_dev on unequal term shards (one of them possibly
    empty), their parts laid out rank after rank as an all-gather would, ecgpu_msm_finish_dev over all of them == the
    one-call MSM == the oracle (sample).  Identities mixed in."""
    ecgpu = ecgpu_module()
    c = pyec.CURVES[curve]
    L = c.L
    k = rand_scalars(c.cid, n, 0xEC0003F7 + c.cid)
    s = rand_scalars(c.cid, n, 0xEC0004F7 + c.cid)
    pts, _ = eng.mul_by_generator(c.cid, s)
    pts = pts.copy()
    inf = np.zeros(n, np.uint8)
    inf[::7] = 1
    pts.reshape(n, 2 * L)[::7] = 0
    want, wf = eng.lincomb(c.cid, k, pts, inf)
    if n <= 5000:
        w, wi = oracle_lib.msm(c.cid, k, pts, inf, vartime=True)
        assert bytes(want) == bytes(w) and wf == wi
    bounds = [ecgpu.shard_range(n, r, shards) for r in range(shards)]
    plan_terms = max(hi - lo for lo, hi in bounds)
    nbytes = eng.msm_parts_bytes(c.cid, plan_terms)
    assert nbytes % 16 == 0 and nbytes > 0
    d_all = eng.dev_alloc(shards * nbytes)
    d_k, d_p, d_i = eng.to_device(k), eng.to_device(pts), eng.to_device(inf)
    pad = lambda x: (x + 15) // 16 * 16
    for r, (lo, hi) in enumerate(bounds):
        # shard inputs copied to 16-byte aligned device buffers of their own (a rank holds only its slice)
        m = hi - lo
        dk = eng.to_device(k[lo * L: hi * L]) if m else None
        dp = eng.to_device(pts[lo * 2 * L: hi * 2 * L]) if m else None
        di = eng.to_device(inf[lo:hi]) if m else None
        eng.msm_parts_dev(c.cid, dk, dp, di, m, plan_terms, d_all.at(r * nbytes))
        for b in (dk, dp, di):
            if b is not None:
                b.free()
    d_o, d_f = eng.dev_alloc(pad(2 * L)), eng.dev_alloc(16)
    eng.msm_finish_dev(c.cid, d_all, shards, plan_terms, d_o, d_f)
    got, gf = eng.to_host(d_o, 2 * L), int(eng.to_host(d_f, 1)[0])
    assert bytes(got) == bytes(want) and gf == wf
    for b in (d_all, d_k, d_p, d_i, d_o, d_f):
        b.free()


@pytest.mark.parametrize("devices,mode", [([0, 0], "peer"), ([0, 0, 0], "peer"), ([0], "rccl"), ([0], "peer")])
def test_group_multi_device_entry_points(eng, devices, mode, monkeypatch):
    """ecgpu_group_*: the single-process multi-GPU entry (one context + worker thread per member).  On a one-GPU box the
    members are contexts on the same device: term shards, the parts exchange by peer copy (or RCCL's all-gather in a
    one-member communicator, which is what can be exercised here), one combining step.  lincomb, mul_by_generator and mul
    must give the bytes of the single-context calls, for a GLV-sized and a tiny MSM, with identities mixed in."""
    ecgpu = ecgpu_module()
    try:
        grp = ecgpu.Group(devices, exchange=mode)          # ecgpu_group_set_exchange: "rccl" = RCCL or an error, "peer" = peer copies
    except ecgpu.EcgpuError:
        if mode == "rccl":
            pytest.skip("librccl could not be loaded in this process")
        raise
    try:
        assert grp.size == len(devices) and grp.exchange == mode
        for curve in ("k256", "p256"):
            c = pyec.CURVES[curve]
            for n in ((1 << 15) + 13, 5, 0):
                k = rand_scalars(c.cid, n, 0xEC0005F7 + c.cid + n)
                s = rand_scalars(c.cid, n, 0xEC0006F7 + c.cid + n)
                pts, _ = eng.mul_by_generator(c.cid, s)
                pts = pts.copy()
                inf = np.zeros(n, np.uint8)
                inf[::5] = 1
                pts.reshape(n, 2 * c.L)[::5] = 0
                want, wf = eng.lincomb(c.cid, k, pts, inf)
                got, gf = grp.lincomb(c.cid, k, pts, inf)
                assert bytes(got) == bytes(want) and gf == wf
                if n:                                           # without the flags the zeroed records are (0, 0): off the curve
                    with pytest.raises(ecgpu.EcgpuError) as ei:
                        grp.lincomb(c.cid, k, pts)
                    assert ei.value.code == ecgpu.ERR_POINT
            n = 3000
            k = rand_scalars(c.cid, n, 0xEC0007F7 + c.cid)
            a, ai = grp.mul_by_generator(c.cid, k)
            b, bi = eng.mul_by_generator(c.cid, k)
            assert bytes(a) == bytes(b) and bytes(ai) == bytes(bi)
            k2 = rand_scalars(c.cid, n, 0xEC0008F7 + c.cid)
            a2, ai2 = grp.mul(c.cid, k2, b)
            b2, bi2 = eng.mul(c.cid, k2, b)
            assert bytes(a2) == bytes(b2) and bytes(ai2) == bytes(bi2)
            # the signature entry points: index-range slices over the members == the single-context calls
            from gpu_common import recover_cases, recover_pack
            z, r, s_, q, exp = ecdsa_pack(ecdsa_cases(c, 0x6E0 + c.cid, nvalid=5))
            assert bytes(grp.ecdsa_verify(c.cid, z, r, s_, q)) == bytes(exp) == bytes(eng.ecdsa_verify(c.cid, z, r, s_, q))
            rz, rr_, rs, rid, rxy, rok = recover_pack(recover_cases(c, 0x6E1 + c.cid, nvalid=5), c.L)
            gk, gv = grp.ecdsa_recover(c.cid, rz, rr_, rs, rid)
            assert bytes(gk) == rxy and bytes(gv) == bytes(rok)
            sg = b"".join(r[i * c.L:(i + 1) * c.L] + s_[i * c.L:(i + 1) * c.L] for i in range(len(exp)))
            msgs = bytes(range(7)) * len(exp)
            assert bytes(grp.ecdsa_verify_msg(c.cid, q, msgs, 7, sg)) == bytes(eng.ecdsa_verify_msg(c.cid, q, msgs, 7, sg))
    finally:
        grp.close()


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_host_pointer_msm_chunked_path_every_curve(eng, keng, curve, monkeypatch):
    """The chunked host-pointer ecgpu_msm on every parameter set, with the chunk size lowered to 2^10 terms
    (ECGPU_MSM_PIPE_LOG2) so that 2^11 + 37 terms take it: the partial-sum records sit at a pitch of 2L bytes, which is not a
    multiple of 16 for p224 (56) and p521 (132).  Result == the unchunked call == the oracle; identities mixed in."""
    c = pyec.CURVES[curve]
    n = (1 << 11) + 37
    k = rand_scalars(c.cid, n, 0xEC0001F7 + c.cid)
    s = rand_scalars(c.cid, n, 0xEC0002F7 + c.cid)
    pts, _ = eng.mul_by_generator(c.cid, s)
    pts = pts.copy()
    inf = np.zeros(n, np.uint8)
    inf[[0, 1023, 1024, n - 1]] = 1
    pts.reshape(n, 2 * c.L)[inf == 1] = 0
    plain, pf = eng.lincomb(c.cid, k, pts, inf)
    monkeypatch.setenv("ECGPU_MSM_PIPE_LOG2", "10")
    o, f = keng.lincomb(c.cid, k, pts, inf)
    monkeypatch.delenv("ECGPU_MSM_PIPE_LOG2")
    assert bytes(o) == bytes(plain) and f == pf
    w, wf = oracle_lib.msm(c.cid, k, pts, inf, vartime=True)
    assert bytes(o) == bytes(w) and f == wf


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_fixed_base_compressed_output(eng, curve):
    """tag || x of ecgpu_batch_mul_base_compressed is the SEC1 compressed encoding of the x || y result: 02 / 03 by the
    parity of y, 00 and x = 0 for the identity; also through the pipelined host path (n >= 2^19)."""
    c = pyec.CURVES[curve]
    for n in (700, (1 << 19) + 77):
        scal = rand_scalars(c.cid, n, 0xEC0000C7 + c.cid).copy()
        scal[: c.L] = 0
        scal[3 * c.L: 4 * c.L] = np.frombuffer(pyec.enc_scalar(c, c.n - 1), np.uint8)
        xy, inf = eng.mul_by_generator(c.cid, scal)
        x, tag = eng.mul_by_generator_compressed(c.cid, scal)
        xy = xy.reshape(n, 2 * c.L)
        assert bytes(x) == bytes(np.ascontiguousarray(xy[:, : c.L]))
        want_tag = np.where(inf == 1, 0, 2 + (xy[:, 2 * c.L - 1] & 1)).astype(np.uint8)
        assert bytes(tag) == bytes(want_tag) and tag[0] == 0 and not x[: c.L].any()
    # against the big-int model's encoding of a few points
    for i in (1, 2, 3):
        k = int.from_bytes(bytes(scal[i * c.L: (i + 1) * c.L]), "big")
        P = pyec.mul(c, k, pyec.G(c))
        assert bytes(x[i * c.L: (i + 1) * c.L]) == P[0].to_bytes(c.L, "big") and tag[i] == 2 + (P[1] & 1)


def test_asynchronous_mode_queues_calls_and_defers_input_errors(eng):
    """ecgpu_set_async: `_dev` calls return once queued, results after ecgpu_synchronize equal the oracle's, an input-check
    error of a queued call surfaces at ecgpu_synchronize (once), host-pointer calls stay synchronous and report only
    their own errors (include/ecgpu.h, 'Asynchronous mode')."""
    ecgpu = ecgpu_module()
    c = pyec.CURVES["k256"]
    L, n = c.L, 3000
    pad = lambda x: (x + 15) // 16 * 16
    ks = [rand_scalars(c.cid, n, 0xA5F0 + i) for i in range(3)]
    pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xA5FF))
    d_k = [eng.to_device(k) for k in ks]
    d_p = eng.to_device(pts)
    d_o = [eng.dev_alloc(pad(n * 2 * L)) for _ in range(4)]
    d_f = [eng.dev_alloc(pad(n)) for _ in range(4)]
    eng.set_async(True)
    try:
        eng.mul_by_generator_dev(c.cid, d_k[0], n, d_o[0], d_f[0])
        eng.mul_by_generator_dev(c.cid, d_k[1], n, d_o[1], d_f[1])
        eng.mul_dev(c.cid, d_k[2], d_p, None, n, d_o[2], d_f[2])
        eng.lincomb_dev(c.cid, d_k[0], d_p, None, n, d_o[3], d_f[3])
        eng.synchronize()
        assert eng.last_timing("total") is not None
        for i in range(2):
            w, wf = oracle_lib.batch_mul_base(c.cid, ks[i])
            assert bytes(eng.to_host(d_o[i], n * 2 * L)) == bytes(w) and bytes(eng.to_host(d_f[i], n)) == bytes(wf)
        w, wf = oracle_lib.batch_mul(c.cid, ks[2], pts)
        assert bytes(eng.to_host(d_o[2], n * 2 * L)) == bytes(w) and bytes(eng.to_host(d_f[2], n)) == bytes(wf)
        w, wf = oracle_lib.msm(c.cid, ks[0], pts, np.zeros(n, np.uint8), vartime=True)
        assert bytes(eng.to_host(d_o[3], 2 * L)) == bytes(w) and int(eng.to_host(d_f[3], 1)[0]) == wf
        # a scalar >= n in a queued batch: the call returns, the error arrives with synchronize, once
        bad = ks[1].copy()
        bad[5 * L: 6 * L] = 0xFF
        d_bad = eng.to_device(bad)
        eng.mul_by_generator_dev(c.cid, d_k[0], n, d_o[0], d_f[0])
        eng.mul_by_generator_dev(c.cid, d_bad, n, d_o[1], d_f[1])
        eng.mul_by_generator_dev(c.cid, d_k[1], n, d_o[2], d_f[2])
        # a host-pointer call in between runs synchronously and does not report the queued call's error
        o, f = eng.mul_by_generator(c.cid, ks[2])
        w, wf = oracle_lib.batch_mul_base(c.cid, ks[2])
        assert bytes(o) == bytes(w) and bytes(f) == bytes(wf)
        with pytest.raises(ecgpu.EcgpuError) as ei:
            eng.synchronize()
        assert ei.value.code == ecgpu.ERR_SCALAR_RANGE
        eng.synchronize()
        w, wf = oracle_lib.batch_mul_base(c.cid, ks[1])          # the batch queued after the bad one is intact
        assert bytes(eng.to_host(d_o[2], n * 2 * L)) == bytes(w)
        # an error of a host-pointer call on the asynchronous context is its own, and immediate
        with pytest.raises(ecgpu.EcgpuError) as ei:
            eng.mul_by_generator(c.cid, bad)
        assert ei.value.code == ecgpu.ERR_SCALAR_RANGE
        eng.synchronize()
        d_bad.free()
    finally:
        eng.set_async(False)
    # back in the synchronous mode errors are immediate again
    d_bad = eng.to_device(bad)
    with pytest.raises(ecgpu.EcgpuError) as ei:
        eng.mul_by_generator_dev(c.cid, d_bad, n, d_o[1], d_f[1])
    assert ei.value.code == ecgpu.ERR_SCALAR_RANGE
    eng.mul_by_generator_dev(c.cid, d_k[0], n, d_o[0], d_f[0])
    for b in d_k + d_o + d_f + [d_p, d_bad]:
        b.free()


def test_device_pointer_entry_points():
    """The *_dev forms on torch tensors (what bench.py and a torch-based caller use) give the same bytes as the
    host-pointer forms: tests/gpu_dev_pointer_check.py, in a process of its own — torch has to be imported before
    libecgpu.so so that both use the HIP runtime torch ships."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "gpu_dev_pointer_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "device-pointer entry points: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_timing_events_can_be_switched_off(eng):
    """ecgpu_set_timing(0): a call puts nothing but its kernels on the stream and ecgpu_last_timing has nothing to report;
    the results do not depend on it; switching it on again brings the stage timings back (include/ecgpu.h)."""
    c = pyec.CURVES["k256"]
    L, n = c.L, 1500
    pad = lambda x: (x + 15) // 16 * 16
    ks = rand_scalars(c.cid, n, 0x71A1)
    want, want_inf = oracle_lib.batch_mul_base(c.cid, ks)
    d_k = eng.to_device(ks)
    d_o, d_f = eng.dev_alloc(pad(n * 2 * L)), eng.dev_alloc(pad(n))
    try:
        eng.set_timing(False)
        eng.mul_by_generator_dev(c.cid, d_k, n, d_o, d_f)
        assert eng.last_timing("main") is None and eng.last_timing("total") is None
        assert bytes(eng.to_host(d_o, n * 2 * L)) == bytes(want) and bytes(eng.to_host(d_f, n)) == bytes(want_inf)
        eng.set_async(True)
        for _ in range(3):
            eng.mul_by_generator_dev(c.cid, d_k, n, d_o, d_f)
        eng.synchronize()
        assert eng.last_timing("main") is None
        eng.set_timing(True)
        eng.mul_by_generator_dev(c.cid, d_k, n, d_o, d_f)
        eng.synchronize()
        assert eng.last_timing("main") > 0 and eng.last_timing("normalize") > 0
        assert bytes(eng.to_host(d_o, n * 2 * L)) == bytes(want)
    finally:
        eng.set_async(False)
        eng.set_timing(True)
        for b in (d_o, d_f, d_k):
            b.free()


@pytest.mark.parametrize("fault,devices", [(1, [0, 0, 0]), (2, [0, 0]), (3, [0, 0]), (3, [0])])
def test_group_exchange_never_hangs_the_caller(eng, fault, devices):
    """The exchange step of ecgpu_group_msm has a deadline (ecgpu_group_set_exchange_timeout).  Fault injection through the exported
    test hook ecgpu_testhook_group_exchange (not in include/ecgpu.h): 1 = the collective's enqueue fails on member 0 while the other
    members' collectives are in flight and can never complete (the partial failure), 2 = it fails everywhere, 3 = it is enqueued
    everywhere and never completes — the way RCCL has actually failed on this pool.  Every time the call must RETURN, with the bytes
    of the single-context MSM, having fallen back to peer copies on fresh streams, say so in exchange_reason, and stay healthy for the
    next call."""
    import ctypes
    import time
    ecgpu = ecgpu_module()
    hook = ecgpu.load_library().ecgpu_testhook_group_exchange
    hook.restype, hook.argtypes = None, [ctypes.c_int]
    c = pyec.CURVES["k256"]
    n = (1 << 15) + 77
    k = rand_scalars(c.cid, n, 0xEC0031F7 + fault)
    pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xEC0032F7 + fault))
    want, wf = eng.lincomb(c.cid, k, pts)
    grp = ecgpu.Group(devices)
    try:
        grp.set_exchange_timeout(1.5)
        hook(fault)
        t0 = time.perf_counter()
        got, gf = grp.lincomb(c.cid, k, pts)
        dt = time.perf_counter() - t0
        hook(0)
        assert bytes(got) == bytes(want) and gf == wf
        assert grp.exchange == "peer" and "ncclAllGather" in grp.exchange_reason, grp.exchange_reason
        if fault == 3:
            assert "did not complete" in grp.exchange_reason and 1.0 < dt < 20.0, (dt, grp.exchange_reason)
        else:
            assert "failed" in grp.exchange_reason and dt < 10.0, (dt, grp.exchange_reason)      # (no wait for the deadline)
        got, gf = grp.lincomb(c.cid, k, pts)                      # the group goes on over peer copies
        assert bytes(got) == bytes(want) and gf == wf
    finally:
        hook(0)
        grp.close()


def test_generator_table_policy_budget_and_pinning():
    """include/ecgpu.h "the generator (comb) tables and their footprint", checked in a process of its own (the tables and the count
    of multiplications seen are per device and process: other tests' contexts would be part of the picture):
    tests/gpu_table_policy_check.py."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_table_policy_check.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("curve", ["k256", "p256", "p384"])
def test_msm_fused_tail_form(eng, keng, curve, monkeypatch):
    """ECGPU_MSM_FUSED_TAIL=1: the bucket finish inside the running sums (k_msm_finish_segments, with the degenerate buckets listed
    by k_msm_find_big before the accumulation and summed by k_msm_big_buckets) — not the default form, kept
    correct: random scalars, and a scalar set that puts thousands of terms into single buckets, against the default form and the
    exact dot product."""
    c = pyec.CURVES[curve]
    n = (1 << 18) + 333                                   # (above every curve's small-MSM threshold: the bucket method)
    s = rand_scalars(c.cid, n, 0xEC0061F7 + c.cid)
    pts, _ = eng.mul_by_generator(c.cid, s)
    for case in ("random", "few distinct scalars"):
        k = rand_scalars(c.cid, n, 0xEC0062F7 + c.cid)
        if case != "random":
            k = np.tile(k[: 3 * c.L], n // 3 + 1)[: n * c.L].copy()          # three scalars: every window has three huge buckets
        want, wf = eng.lincomb(c.cid, k, pts)
        monkeypatch.setenv("ECGPU_MSM_FUSED_TAIL", "1")
        got, gf = keng.lincomb(c.cid, k, pts)
        monkeypatch.delenv("ECGPU_MSM_FUSED_TAIL")
        assert bytes(got) == bytes(want) and gf == wf, (curve, case)
        ki = [int.from_bytes(bytes(k[i * c.L:(i + 1) * c.L]), "big") for i in range(n)]
        si = [int.from_bytes(bytes(s[i * c.L:(i + 1) * c.L]), "big") for i in range(n)]
        dot = sum(a * b for a, b in zip(ki, si)) % c.n
        o, oi = oracle_lib.batch_mul_base(c.cid, np.frombuffer(dot.to_bytes(c.L, "big"), np.uint8))
        assert bytes(got) == bytes(o) and gf == int(oi[0]), (curve, case)


@pytest.mark.parametrize("curve", ["k256", "p256", "p384"])
def test_signature_batches_share_their_scalar_inversions(eng, curve):
    """ECDSA verification and public-key recovery invert one scalar per signature modulo the group order (`Scalar::invert`,
    k256/src/arithmetic/scalar.rs:139-143); k_scalar_batch_inv does it for the whole batch by Montgomery's trick, several signatures per
    lane from 65,537 signatures on.  200,000 signatures (a tile of model-made cases of prime length, so that the signatures of one lane
    are different cases; range failures — s = 0, r = 0, s = n — among them, which must drop out of the lane's product): every verdict
    and every recovered key must equal the small-batch result (one inversion per lane), which the other tests pin to the oracle."""
    from gpu_common import recover_cases, recover_pack
    c = pyec.CURVES[curve]
    L = c.L
    z, r, s, q, exp = ecdsa_pack(ecdsa_cases(c, 0x1B1 + c.cid, nvalid=24))
    m = len(exp)
    small = eng.ecdsa_verify(c.cid, z, r, s, q)
    assert bytes(small) == bytes(exp)
    n = 200003                                        # (prime: the lane stride ceil(n / 4) is no multiple of the tile length)
    reps = n // m + 1
    big = eng.ecdsa_verify(c.cid, (z * reps)[: n * L], (r * reps)[: n * L], (s * reps)[: n * L], (q * reps)[: n * 2 * L])
    assert bytes(big) == bytes(np.tile(exp, reps)[:n])
    rz, rr, rs, rid, rxy, rok = recover_pack(recover_cases(c, 0x1B2 + c.cid, nvalid=12), L)
    m2 = len(rok)
    k_small, ok_small = eng.ecdsa_recover(c.cid, rz, rr, rs, rid)
    assert bytes(k_small) == rxy and bytes(ok_small) == bytes(rok)
    n = 140009
    reps = n // m2 + 1
    k_big, ok_big = eng.ecdsa_recover(c.cid, (rz * reps)[: n * L], (rr * reps)[: n * L], (rs * reps)[: n * L], (bytes(rid) * reps)[:n])
    assert bytes(k_big) == (rxy * reps)[: n * 2 * L] and bytes(ok_big) == (bytes(rok) * reps)[:n]


@pytest.mark.parametrize("curve,n", [("k256", (1 << 17) + 11), ("k256", 1 << 21), ("p256", (1 << 16) + 3)])
def test_sharded_msm_steps_on_rotating_lanes(curve, n):
    """The throughput form of the sharded MSM (include/ecgpu.h, ecgpu_msm_parts_join_dev): on an asynchronous context with two MSM
    lanes the local halves of consecutive MSMs run on rotating internal streams; the caller joins the record it is about to exchange
    and queues the combining half on the context's stream, beside the next local half.  Five different MSMs in a software pipeline
    (local half of i, THEN join + combining half of i - 1; two "ranks" = two shards per MSM, both computed here) must give the bytes
    of five one-call MSMs; a bad point in step 3 surfaces at ecgpu_synchronize.  k256 at 2^21 terms: the plain-scalar plan."""
    ecgpu = ecgpu_module()
    e = ecgpu.Engine(0)
    try:
        c = pyec.CURVES[curve]
        L = c.L
        steps, shards, lanes = 5, 2, 2
        s = rand_scalars(c.cid, n, 0xEC0071F7 + c.cid)
        pts, _ = e.mul_by_generator(c.cid, s)
        ks = [rand_scalars(c.cid, n, 0xEC0072F7 + c.cid + i) for i in range(steps)]
        want = [e.lincomb(c.cid, k, pts) for k in ks]
        bounds = [ecgpu.shard_range(n, r, shards) for r in range(shards)]
        plan_terms = max(hi - lo for lo, hi in bounds)
        nbytes = e.msm_parts_bytes(c.cid, plan_terms)
        d_p = [e.to_device(pts[lo * 2 * L: hi * 2 * L]) for lo, hi in bounds]
        d_k = [[e.to_device(k[lo * L: hi * L]) for lo, hi in bounds] for k in ks]
        # one gathered record (shards * nbytes) per lane SLOT; an MSM's shards are local halves of their own, so an MSM takes `shards`
        # consecutive lanes turns: with two lanes and two shards, shard r of every MSM runs on lane r
        d_all = [e.dev_alloc(shards * nbytes) for _ in range(lanes)]
        d_o = [e.dev_alloc(2 * L + 32) for _ in range(steps)]
        e.set_async(True)
        e.set_msm_lanes(lanes)
        pend = []

        def combine():
            i = pend.pop(0)
            buf = d_all[i % lanes]
            for r in range(shards):
                e.msm_parts_join_dev(buf.at(r * nbytes))
            e.msm_finish_dev(c.cid, buf, shards, plan_terms, d_o[i].at(0), d_o[i].at((2 * L + 15) // 16 * 16))

        for i in range(steps):
            buf = d_all[i % lanes]
            for r, (lo, hi) in enumerate(bounds):
                e.msm_parts_dev(c.cid, d_k[i][r], d_p[r], None, hi - lo, plan_terms, buf.at(r * nbytes))
            pend.append(i)
            if len(pend) > 1:
                combine()
        while pend:
            combine()
        e.synchronize()
        for i in range(steps):
            rec = e.to_host(d_o[i], 2 * L + 32)
            assert bytes(rec[: 2 * L]) == bytes(want[i][0]) and int(rec[(2 * L + 15) // 16 * 16]) == want[i][1], (curve, n, i)
        # an input error in a queued local half: deferred to ecgpu_synchronize, exactly once
        bad = pts[: 2 * L * (bounds[0][1] - bounds[0][0])].copy()
        bad[2 * L - 1] ^= 1
        d_bad = e.to_device(bad)
        e.msm_parts_dev(c.cid, d_k[0][0], d_bad, None, bounds[0][1] - bounds[0][0], plan_terms, d_all[0].at(0))
        with pytest.raises(ecgpu.EcgpuError) as ei:
            e.synchronize()
        assert ei.value.code == ecgpu.ERR_POINT
        e.synchronize()
        e.set_msm_lanes(1)
        e.set_async(False)
    finally:
        e.close()


def test_group_exchange_choice_is_an_api_call_not_an_environment_variable(monkeypatch):
    """ecgpu_group_set_exchange: a group of duplicate devices has no RCCL exchange — asking for "RCCL or nothing" is an error that says
    why, asking for peer copies is fine; and the product library ignores ECGPU_GROUP_EXCHANGE (round 5 read it)."""
    ecgpu = ecgpu_module()
    monkeypatch.setenv("ECGPU_GROUP_EXCHANGE", "rccl")
    grp = ecgpu.Group([0, 0])                        # (round 5: ECGPU_ERR_HIP with this variable set)
    try:
        assert grp.exchange == "peer" and "duplicate devices" in grp.exchange_reason
        with pytest.raises(ecgpu.EcgpuError) as ei:
            grp.set_exchange("rccl")
        assert ei.value.code == ecgpu.ERR_HIP and "no RCCL exchange" in str(ei.value)
        grp.set_exchange("peer")
        assert grp.exchange == "peer"
    finally:
        grp.close()
    monkeypatch.setenv("ECGPU_GROUP_EXCHANGE", "peer")
    try:
        grp = ecgpu.Group([0])
    except ecgpu.EcgpuError:
        pytest.skip("no group on this box")
    try:
        if grp.exchange == "rccl":                   # (librccl loadable: the variable did not turn it off)
            grp.set_exchange("peer")
            assert grp.exchange == "peer" and "ecgpu_group_set_exchange" in grp.exchange_reason
    finally:
        grp.close()
