"""-m gpu: device-side known-answer tests of the field and group arithmetic (ecgpu_selftest_field / ecgpu_selftest_point:
the __host__ __device__ code of ecgpu_field.h / ecgpu_point.h running as gfx950 code, one lane per element) against the
reference's field vectors, integers mod p, the oracle and the big-integer group model."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib
import pyec
from gpu_common import ALL_CURVES, ecgpu_module

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def eng():
    e = ecgpu_module().Engine(0)
    oracle_lib.build()
    yield e
    e.close()


def fe(c, vals):
    return np.frombuffer(b"".join(v.to_bytes(c.L, "big") for v in vals), np.uint8)


def ints(c, arr):
    b = bytes(arr)
    return [int.from_bytes(b[i: i + c.L], "big") for i in range(0, len(b), c.L)]


def edge_values(c):
    return [0, 1, 2, 3, c.p - 1, c.p - 2, (c.p - 1) // 2, (c.p + 1) // 2, 2 ** 32 - 1, 2 ** 32, 2 ** 64 - 1,
            2 ** (8 * c.L - 1) % c.p, (2 ** (8 * c.L) - 1) % c.p, 0x1000003D1 % c.p, c.p - 0x1000003D1,
            int("ff" * c.L, 16) % c.p, int("80" + "00" * (c.L - 1), 16) % c.p]


@pytest.mark.parametrize("curve", ["k256", "p256"])
def test_device_field_doubling_vectors_of_the_reference(eng, curve):
    """k256/src/test_vectors/field.rs (DBL_TEST_VECTORS, used at k256/src/arithmetic/field.rs tests) and
    p256/src/arithmetic/field.rs:219-245: repeated doubling of 1, every step on the device (a + a and 2a)."""
    c = pyec.CURVES[curve]
    with open(os.path.join(GOLDEN, curve + ".json")) as f:
        vec = [int(v, 16) for v in json.load(f)["field_dbl"]]
    a = fe(c, vec[:-1])
    assert ints(c, eng.selftest_field(c.cid, 0, a, a)) == vec[1:]
    assert ints(c, eng.selftest_field(c.cid, 7, a)) == vec[1:]


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_device_field_ops_vs_bigint_and_oracle(eng, curve):
    c = pyec.CURVES[curve]
    rng = random.Random(0xD0F00 + c.cid)
    vals = edge_values(c) + [rng.randrange(c.p) for _ in range(3000)]
    other = [vals[(i * 5 + 1) % len(vals)] for i in range(len(vals))]
    A, B = fe(c, vals), fe(c, other)
    p = c.p
    want = {0: [(a + b) % p for a, b in zip(vals, other)], 1: [(a - b) % p for a, b in zip(vals, other)],
            2: [a * b % p for a, b in zip(vals, other)], 8: [(2 * a + b) % p for a, b in zip(vals, other)],
            9: [(-b * b) % p for a, b in zip(vals, other)],
            13: [(a * b - 2 * a - b) % p for a, b in zip(vals, other)],          # a*b - c in one reduction (k256: F::mul_sub)
            14: [((a + b) ** 2 - 5 * b) % p for a, b in zip(vals, other)]}       # a^2 - c (k256: F::sqr_sub)
    for op, w in want.items():
        assert ints(c, eng.selftest_field(c.cid, op, A, B)) == w, (curve, op)
    if curve == "k256":
        # the reduction in assembly (csrc/ecgpu_k256_reduce_asm.h) against the compiler's rendering of the same function, from the same
        # product columns, on lazy operands at the largest limb magnitudes (7 x 1, and 3 x 2 + 1 x 1 under one reduction): the device
        # compares the nine limbs (a mismatch comes back as all-ones bytes), the value is checked here
        adv = edge_values(c) + [p - 1 - k for k in range(40)] + [(1 << 256) - 1 - (1 << 33) - (k << 40) for k in range(8)] + vals[:2000]
        advo = [adv[(i * 7 + 3) % len(adv)] for i in range(len(adv))]
        assert ints(c, eng.selftest_field(c.cid, 15, fe(c, adv), fe(c, advo))) == [14 * a * b % p for a, b in zip(adv, advo)]
    assert ints(c, eng.selftest_field(c.cid, 3, A)) == [a * a % p for a in vals]
    assert ints(c, eng.selftest_field(c.cid, 5, A)) == [(-a) % p for a in vals]
    assert ints(c, eng.selftest_field(c.cid, 7, A)) == [2 * a % p for a in vals]
    inv = [pow(a, -1, p) if a else 0 for a in vals]
    assert ints(c, eng.selftest_field(c.cid, 4, A)) == inv                     # safegcd division steps
    assert ints(c, eng.selftest_field(c.cid, 17, A)) == inv                    # ... in their variable-time form (ModInv::invert_var)
    assert ints(c, eng.selftest_field(c.cid, 10, A[: 64 * c.L])) == inv[:64]   # Fermat chain
    # the oracle on a sample (it restates the reference's own field code)
    for i in range(0, len(vals), 97):
        a, b = vals[i].to_bytes(c.L, "big"), other[i].to_bytes(c.L, "big")
        for op in (0, 1, 2):
            got = bytes(eng.selftest_field(c.cid, op, A[i * c.L: (i + 1) * c.L], B[i * c.L: (i + 1) * c.L]))
            assert got == oracle_lib.field_op(c.cid, op, a, b)
    # the lazily reduced 25-step chain, against the same chain on integers
    def chain(x, y):
        for _ in range(25):
            t = x * y % p
            u = (t + x - 2 * y) % p
            v = (-(t + 4 * y)) % p
            x, y = u * u % p, v
        return (x + y) % p
    assert ints(c, eng.selftest_field(c.cid, 12, A[: 200 * c.L], B[: 200 * c.L])) == [chain(a, b) for a, b in zip(vals[:200], other[:200])]
    got = ints(c, eng.selftest_field(c.cid, 11, A[: 300 * c.L]))                # sqrt ((p + 1) / 4; p224: Tonelli-Shanks) or 0
    for a, g in zip(vals[:300], got):
        r = pyec.sqrt_mod(a, p)
        assert (g in (r, p - r) and g * g % p == a) if r is not None else g == 0
    ecgpu = ecgpu_module()
    with pytest.raises(ecgpu.EcgpuError) as e:                                   # non-canonical input
        eng.selftest_field(c.cid, 0, fe(c, [1]), np.frombuffer(p.to_bytes(c.L, "big"), np.uint8))
    assert e.value.code == ecgpu.ERR_POINT


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_device_point_ops_vs_model_and_oracle(eng, curve):
    """Complete formulas on every edge pair (P + P, P - P, O + P, P + O, O + O) and the incomplete Jacobian / XYZZ formulas
    of the ladders, the comb and the bucket sums inside their domain, all on the device."""
    c = pyec.CURVES[curve]
    rng = random.Random(0xD9017 + c.cid)
    G = pyec.G(c)
    base = [G, pyec.neg(c, G), pyec.mul(c, 2, G), pyec.INF] + [pyec.mul(c, rng.randrange(1, c.n), G) for _ in range(20)]
    P = [a for a in base for _ in base]
    Q = [b for _ in base for b in base]
    enc = lambda pts: (np.frombuffer(b"".join(pyec.enc_point(c, x)[0] for x in pts), np.uint8),
                       np.array([pyec.enc_point(c, x)[1] for x in pts], np.uint8))
    pxy, pinf = enc(P)
    qxy, qinf = enc(Q)
    dec = lambda out, inf: [pyec.dec_point(c, bytes(out[2 * c.L * i: 2 * c.L * (i + 1)]), int(inf[i])) for i in range(len(inf))]
    add = [pyec.add(c, a, b) for a, b in zip(P, Q)]
    sub = [pyec.add(c, a, pyec.neg(c, b)) for a, b in zip(P, Q)]
    for op, want in ((0, add), (1, add), (4, sub), (5, sub)):
        assert dec(*eng.selftest_point(c.cid, op, pxy, pinf, qxy, qinf)) == want, (curve, op)
    assert dec(*eng.selftest_point(c.cid, 2, pxy, pinf)) == [pyec.add(c, a, a) for a in P]
    assert dec(*eng.selftest_point(c.cid, 3, pxy, pinf)) == [pyec.neg(c, a) for a in P]
    # oracle (the reference's complete addition restated) on the same pairs, bit for bit
    out, inf = eng.selftest_point(c.cid, 0, pxy, pinf, qxy, qinf)
    for i in range(0, len(P), 7):
        w, wi = oracle_lib.point_op(c.cid, 0, bytes(pxy[2 * c.L * i: 2 * c.L * (i + 1)]), int(pinf[i]), bytes(qxy[2 * c.L * i: 2 * c.L * (i + 1)]), int(qinf[i]))
        assert bytes(out[2 * c.L * i: 2 * c.L * (i + 1)]) == w and int(inf[i]) == wi
    # incomplete formulas: finite P, Q with P != +-Q (and 2P != +-Q for the doubling + addition)
    ok = [i for i, (a, b) in enumerate(zip(P, Q)) if a is not pyec.INF and b is not pyec.INF and a != b and a != pyec.neg(c, b)
          and pyec.add(c, a, a) not in (b, pyec.neg(c, b))]
    assert len(ok) > 300
    sel = lambda arr, w: np.concatenate([arr[w * i: w * (i + 1)] for i in ok])
    fp, fq = sel(pxy, 2 * c.L), sel(qxy, 2 * c.L)
    Pf, Qf = [P[i] for i in ok], [Q[i] for i in ok]
    assert dec(*eng.selftest_point(c.cid, 6, fp)) == [pyec.add(c, a, a) for a in Pf]
    assert dec(*eng.selftest_point(c.cid, 7, fp, None, fq)) == [pyec.add(c, pyec.add(c, a, a), b) for a, b in zip(Pf, Qf)]
    for op in (8, 9):
        assert dec(*eng.selftest_point(c.cid, op, fp, None, fq)) == [pyec.add(c, a, b) for a, b in zip(Pf, Qf)], (curve, op)


def test_device_row_parallel_k256_field_and_doubling(eng):
    """csrc/ecgpu_rows.h — the k256 field spread over rows of 16 lanes (one limb per lane, four products per wave; model:
    tools/rows_field_model.py), which k_msm_combine's Horner chain runs its doublings on.  Field op 16: lane i gets 7 a b + 3 a 2 b
    = 13 a b of lane (i mod 4) of its wave — products at limb magnitudes 7 x 1 and 3 x 2, edge values in the first four lanes of
    some wave.  Point op 10: every lane gets 32 P of its wave's first point, by five row-parallel complete doublings (the identity
    and points of every kind in first place)."""
    c = pyec.CURVES["k256"]
    p = c.p
    rng = random.Random(0x50775)
    edge = edge_values(c) + [p - 1 - k for k in range(19)] + [(1 << 256) - 1 - (1 << 33) - (k << 40) for k in range(8)]
    nw = 96                                                    # waves
    vals, other = [], []
    for w in range(nw):
        for l in range(64):
            if l < 4 and w < 40:
                a, b = edge[(4 * w + l) % len(edge)], edge[(7 * w + 3 * l + 1) % len(edge)]
            else:
                a, b = rng.randrange(p), rng.randrange(p)
            vals.append(a)
            other.append(b)
    got = ints(c, eng.selftest_field(c.cid, 16, fe(c, vals), fe(c, other)))
    want = [13 * vals[i - i % 64 + i % 4] * other[i - i % 64 + i % 4] % p for i in range(len(vals))]
    assert got == want
    G = pyec.G(c)
    firsts = [pyec.INF, G, pyec.neg(c, G), pyec.mul(c, 2, G), pyec.mul(c, (c.n - 1) // 2, G), pyec.mul(c, (c.n + 1) // 2, G)] + \
             [pyec.mul(c, rng.randrange(1, c.n), G) for _ in range(10)]
    P = []
    for f in firsts:
        P += [f] + [pyec.mul(c, rng.randrange(1, c.n), G) for _ in range(3)] + [G] * 60     # the other lanes' points must not matter
    pxy = np.frombuffer(b"".join(pyec.enc_point(c, x)[0] for x in P), np.uint8)
    pinf = np.array([pyec.enc_point(c, x)[1] for x in P], np.uint8)
    out, inf = eng.selftest_point(c.cid, 10, pxy, pinf)
    got = [pyec.dec_point(c, bytes(out[2 * c.L * i: 2 * c.L * (i + 1)]), int(inf[i])) for i in range(len(P))]
    want = []
    for f in firsts:
        want += [pyec.mul(c, 32, f) if f is not pyec.INF else pyec.INF] * 64
    assert got == want
