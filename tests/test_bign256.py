"""bign-curve256v1 (`bignp256` in the reference): the twelfth parameter set, a = -3 on the any-a formulas, generator (0, y),
LITTLE-endian wire records (bignp256/src/lib.rs:102).  Pinned by the reference's own group vectors
(bignp256/src/test_vectors/group.rs -> tests/golden/bign256.json).  CPU tests: the oracle and the host build of the
kernels' arithmetic; -m gpu tests: the HIP path through the C ABI."""
import json
import os
import random

import numpy as np
import pytest

import hostcheck_lib as hc
import oracle_lib
import pyec

C = pyec.CURVES["bign256"]
CID = C.cid
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bign256.json")


def vectors():
    with open(GOLDEN) as f:
        g = json.load(f)["group"]
    ks = [pyec.enc_scalar(C, v["k"]) for v in g["add"]] + [bytes.fromhex(v["k"]) for v in g["mul"]]
    want = b"".join(bytes.fromhex(v["x"]) + bytes.fromhex(v["y"]) for v in g["add"] + g["mul"])
    return ks, want


def rand_scalars(rng, n):
    edge = [0, 1, 2, C.n - 1, C.n - 2, (C.n - 1) // 2, 1 << 128, (1 << 255) % C.n]
    return (edge + [rng.randrange(C.n) for _ in range(n)])[:max(n, len(edge))]


def test_constants_and_wire_order():
    """The generator is (0, y) and the records are little-endian: the first ADD vector is G itself."""
    ks, want = vectors()
    assert pyec.on_curve(C, pyec.G(C)) and C.gx == 0
    assert pyec.enc_point(C, pyec.G(C))[0] == want[:64]
    assert ks[1] == (2).to_bytes(32, "little")
    assert oracle_lib.FIELD_BYTES[CID] == 32 and oracle_lib.CURVE_IDS["bign256"] == CID


def test_oracle_reference_group_vectors(oracle):
    """bignp256/src/test_vectors/group.rs through every driver of the oracle (primeorder/src/dev.rs:100-150 shape):
    mul_by_generator, mul (constant-time LUT path), mul_vartime (wNAF), repeated addition; and the big-int model."""
    ks, want = vectors()
    n = len(ks)
    scal = b"".join(ks)
    gxy = pyec.enc_point(C, pyec.G(C))[0] * n
    for out, inf in (oracle.batch_mul_base(CID, scal), oracle.batch_mul(CID, scal, gxy), oracle.batch_mul(CID, scal, gxy, vartime=True)):
        assert bytes(out) == want and not inf.any()
    cur = want[:64]
    for i in range(1, 20):                                   # G + G + ... (full and mixed addition)
        for op in (0, 1):
            nxt, inf = oracle.point_op(CID, op, cur, 0, want[:64], 0)
            assert inf == 0 and nxt == want[64 * i: 64 * i + 64]
        cur = nxt
    for k, i in zip(ks, range(n)):
        assert pyec.dec_point(C, want[64 * i: 64 * i + 64], 0) == pyec.mul(C, int.from_bytes(k, "little"), pyec.G(C))


def test_oracle_vs_model_variable_base_and_lincomb(oracle):
    rng = random.Random(0xB167)
    G = pyec.G(C)
    pts = [G, pyec.neg(C, G), pyec.INF] + [pyec.mul(C, rng.randrange(1, C.n), G) for _ in range(9)]
    ks = rand_scalars(rng, 12)
    enc = [pyec.enc_point(C, P) for P in pts]
    scal = b"".join(pyec.enc_scalar(C, k) for k in ks)
    pxy, pinf = b"".join(e[0] for e in enc), np.array([e[1] for e in enc], np.uint8)
    out, inf = oracle.batch_mul(CID, scal, pxy, pinf)
    for i in range(12):
        assert pyec.dec_point(C, bytes(out[64 * i: 64 * i + 64]), int(inf[i])) == pyec.mul(C, ks[i], pts[i])
    for vt in (False, True):
        o, f = oracle.msm(CID, scal, pxy, pinf, vartime=vt)
        assert pyec.dec_point(C, bytes(o), f) == pyec.msm(C, ks, pts)
    assert oracle.scalar_reduce(CID, np.frombuffer(C.n.to_bytes(32, "little"), np.uint8)).tolist() == [0] * 32
    with pytest.raises(oracle.OracleError):
        oracle.batch_mul_base(CID, C.n.to_bytes(32, "little"))          # scalar = n: out of range in the curve's own byte order


def test_hostcheck_kernel_arithmetic_vs_oracle(oracle):
    """The kernels' host build (tests/hostcheck): comb, ladder and Pippenger control flow for the twelfth set."""
    rng = random.Random(0xB168)
    ks = rand_scalars(rng, 30)
    scal = b"".join(pyec.enc_scalar(C, k) for k in ks)
    want, winf = oracle.batch_mul_base(CID, scal)
    for w in (4, 13):
        rc, out, inf = hc.batch_mul_base(CID, w, scal, 3)
        assert rc == 0 and bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    pts = want.copy()
    k2 = b"".join(pyec.enc_scalar(C, k) for k in reversed(ks))
    rc, out, inf = hc.batch_mul(CID, k2, pts, winf, nthreads=4)
    w2, wi2 = oracle.batch_mul(CID, k2, pts, winf)
    assert rc == 0 and bytes(out) == bytes(w2) and bytes(inf) == bytes(wi2)
    wm, wf = oracle.msm(CID, k2, pts, winf)
    for cbits, chunk in ((4, 1), (9, 16)):
        rc, out, inf = hc.msm(CID, cbits, k2, pts, winf, chunk=chunk)
        assert rc == 0 and out == bytes(wm) and inf == wf
    a, b = rng.randrange(C.p), rng.randrange(C.p)
    A, B = a.to_bytes(32, "little"), b.to_bytes(32, "little")
    assert int.from_bytes(hc.field_op(CID, 2, A, B), "little") == a * b % C.p
    assert hc.field_op(CID, 2, A, B) == oracle.field_op(CID, 2, A, B)
    assert int.from_bytes(hc.field_op(CID, 4, A), "little") == pow(a, -1, C.p)


# ---------------------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def eng():
    import importlib
    e = importlib.import_module("elliptic-curves_amd").Engine(0)
    oracle_lib.build()
    yield e
    e.close()


@pytest.mark.gpu
def test_gpu_reference_group_vectors(eng):
    """bignp256/src/test_vectors/group.rs through ecgpu_batch_mul_base, ecgpu_batch_mul, ecgpu_msm and (repeated addition)
    ecgpu_point_sum."""
    ks, want = vectors()
    n = len(ks)
    scal = b"".join(ks)
    out, inf = eng.mul_by_generator(CID, scal)
    assert bytes(out) == want and not inf.any()
    gxy = pyec.enc_point(C, pyec.G(C))[0]
    out, inf = eng.mul(CID, scal, gxy * n)
    assert bytes(out) == want and not inf.any()
    for i in range(n):
        o, f = eng.lincomb(CID, ks[i], gxy)
        assert bytes(o) == want[64 * i: 64 * i + 64] and f == 0
    for i in range(1, 21):
        o, f = eng.point_sum(CID, gxy * i)
        assert bytes(o) == want[64 * (i - 1): 64 * i] and f == 0


@pytest.mark.gpu
def test_gpu_vs_oracle(eng):
    """Fixed base (several comb widths), variable base, MSM (per-term path and the bucket method), normalisation,
    decompression, ECDH, compressed output and the device field self-tests against the oracle, little-endian throughout."""
    import importlib
    ecgpu = importlib.import_module("elliptic-curves_amd")
    rng = random.Random(0xB169)
    ks = rand_scalars(rng, 700)
    scal = np.frombuffer(b"".join(pyec.enc_scalar(C, k) for k in ks), np.uint8)
    want, winf = oracle_lib.batch_mul_base(CID, scal)
    for w in (24, 13, 5):
        eng.set_base_window(CID, w)
        out, inf = eng.mul_by_generator(CID, scal)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    eng.set_base_window(CID, 24)
    n = len(ks)
    k2 = np.frombuffer(b"".join(pyec.enc_scalar(C, k) for k in reversed(ks)), np.uint8)
    out, inf = eng.mul(CID, k2, want, winf)
    w2, wi2 = oracle_lib.batch_mul(CID, k2, want, winf)
    assert bytes(out) == bytes(w2) and bytes(inf) == bytes(wi2)
    for m in (0, 1, 17, n):
        for cbits in (0, 9):
            eng.set_msm_window(cbits)
            o, f = eng.lincomb(CID, k2[: 32 * m], want[: 64 * m], winf[:m])
            wm, wf = oracle_lib.msm(CID, k2[: 32 * m], want[: 64 * m], winf[:m], vartime=True)
            assert bytes(o) == bytes(wm) and f == wf, (m, cbits)
    eng.set_msm_window(0)
    x, ok = eng.ecdh(CID, k2[32:], want[64:])                          # skip the identity at index 0 (k = 0)
    assert bytes(x) == bytes(w2.reshape(n, 64)[1:, :32].reshape(-1)) and bytes(ok) == bytes(1 - wi2[1:])
    xs = want.reshape(n, 64)[1:, :32].copy().reshape(-1)
    odd = (want.reshape(n, 64)[1:, 32] & 1).copy()                       # little-endian: the parity is in the FIRST byte of y
    dxy, dok = eng.decompress(CID, xs, odd)
    oxy, ook = oracle_lib.batch_decompress(CID, xs, odd)
    assert bytes(dxy) == bytes(oxy) and bytes(dok) == bytes(ook) and bytes(dxy) == bytes(want[64:])
    cx, tag = eng.mul_by_generator_compressed(CID, scal)
    assert bytes(cx[32:]) == bytes(xs) and bytes(tag[1:]) == bytes(2 + odd) and tag[0] == 0
    # round 4: the constant-time `lincomb` and compressed points into the path, little-endian records
    for m in (0, 1, 3, 17, n):
        o, f = eng.lincomb_ct(CID, k2[: 32 * m], want[: 64 * m], winf[:m])
        wm, wf = oracle_lib.msm(CID, k2[: 32 * m], want[: 64 * m], winf[:m], vartime=False) if m else (np.zeros(64, np.uint8), 1)
        assert bytes(o) == bytes(wm) and f == wf, m
    tags = np.concatenate([np.zeros(1, np.uint8), (2 + odd).astype(np.uint8)])          # index 0 is the identity (k = 0): tag 0, x = 0
    xs_all = np.concatenate([np.zeros(32, np.uint8), xs])
    o, f = eng.lincomb_compressed(CID, k2, xs_all, tags)
    wm, wf = oracle_lib.msm(CID, k2, want, winf, vartime=True)
    assert bytes(o) == bytes(wm) and f == wf
    out, inf = eng.mul_compressed(CID, k2, xs_all, tags)
    assert bytes(out) == bytes(w2) and bytes(inf) == bytes(wi2)
    # device field arithmetic, little-endian records
    vals = [rng.randrange(C.p) for _ in range(500)]
    oth = [rng.randrange(C.p) for _ in range(500)]
    le = lambda v: np.frombuffer(b"".join(x.to_bytes(32, "little") for x in v), np.uint8)
    got = bytes(eng.selftest_field(CID, 2, le(vals), le(oth)))
    assert [int.from_bytes(got[32 * i: 32 * i + 32], "little") for i in range(500)] == [a * b % C.p for a, b in zip(vals, oth)]
    got = bytes(eng.selftest_field(CID, 4, le(vals)))
    assert [int.from_bytes(got[32 * i: 32 * i + 32], "little") for i in range(500)] == [pow(a, -1, C.p) for a in vals]
    # errors in the curve's own byte order
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.mul_by_generator(CID, C.n.to_bytes(32, "little"))
    assert e.value.code == ecgpu.ERR_SCALAR_RANGE
    eng.mul_by_generator(CID, (C.n - 1).to_bytes(32, "little"))
    with pytest.raises(ecgpu.EcgpuError) as e:
        eng.ecdsa_verify(CID, bytes(32), bytes(32), bytes(32), bytes(64))
    assert e.value.code == ecgpu.ERR_CURVE


@pytest.mark.gpu
def test_gpu_large_msm_identity(eng):
    """2^17 + 5 terms (bucket method, two-level sort): sum k_i (s_i G) == (sum k_i s_i) G."""
    rng = np.random.default_rng(0xB16A)
    n = (1 << 17) + 5
    k = oracle_lib.scalar_reduce(CID, rng.integers(0, 256, n * 32, dtype=np.uint8))
    s = oracle_lib.scalar_reduce(CID, rng.integers(0, 256, n * 32, dtype=np.uint8))
    pts, _ = eng.mul_by_generator(CID, s)
    o, f = eng.lincomb(CID, k, pts)
    kb, sb = bytes(k), bytes(s)
    acc = sum(int.from_bytes(kb[32 * i: 32 * i + 32], "little") * int.from_bytes(sb[32 * i: 32 * i + 32], "little") for i in range(n)) % C.n
    w, wf = oracle_lib.batch_mul_base(CID, pyec.enc_scalar(C, acc))
    assert bytes(o) == bytes(w) and f == int(wf[0])
