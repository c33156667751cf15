"""Pins the CPU oracle (oracle/, a C restatement of the reference's algorithms) against every
golden vector the reference holds for the scalar-mul path, and against an independent big-int
model.  CPU only.  (reference tests cited per case.)"""
import json
import os
import random

import numpy as np
import pytest

import pyec

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CURVES = ["k256", "p256", "p384", "p224", "p192", "p521"]          # curves with reference KATs (tests/golden/<curve>.json)
ALL_CURVES = CURVES + ["sm2", "bp256", "bp384", "bp256t1", "bp384t1"]               # + the SURVEY 8(f) rank-4 parameter set: big-int model and OpenSSL only


def load(curve):
    with open(os.path.join(GOLDEN, curve + ".json")) as f:
        return json.load(f)


def xy(c, v):
    return bytes.fromhex(v["x"]) + bytes.fromhex(v["y"])


@pytest.mark.parametrize("curve", CURVES)
def test_add_vectors_repeated_addition(oracle, curve):
    """primeorder/src/dev.rs:20-55 / k256 projective.rs:977-1010: G + G + ... through add,
    mixed add, and double."""
    c = pyec.CURVES[curve]
    vec = load(curve)["group"]["add"]
    g_xy, _ = pyec.enc_point(c, pyec.G(c))
    assert xy(c, vec[0]) == g_xy
    for op in (0, 1):  # full add, mixed add
        cur = g_xy
        for v in vec[1:]:
            cur, inf = oracle.point_op(c.cid, op, cur, 0, g_xy, 0)
            assert inf == 0 and cur == xy(c, v)
    # doubling: 2G, 4G, 8G, 16G are vectors 2, 4, 8, 16
    cur = g_xy
    for k in (2, 4, 8, 16):
        cur, inf = oracle.point_op(c.cid, 2, cur, 0)
        assert cur == xy(c, vec[k - 1])


@pytest.mark.parametrize("curve", CURVES)
def test_mul_vectors_all_drivers(oracle, curve):
    """k256 projective.rs:1040-1087 / primeorder dev.rs:100-150: G * k for ADD (k=1..20) and MUL
    vectors, through mul (const-time LUT path), mul_vartime (wNAF), mul_by_generator (table)."""
    c = pyec.CURVES[curve]
    g = load(curve)["group"]
    ks = [pyec.enc_scalar(c, v["k"]) for v in g["add"]] + [bytes.fromhex(v["k"]) for v in g["mul"]]
    want = b"".join(xy(c, v) for v in g["add"] + g["mul"])
    n = len(ks)
    scal = b"".join(ks)
    gxy = pyec.enc_point(c, pyec.G(c))[0] * n
    for out, inf in (oracle.batch_mul_base(c.cid, scal), oracle.batch_mul(c.cid, scal, gxy),
                     oracle.batch_mul(c.cid, scal, gxy, vartime=True)):
        assert bytes(out) == want
        assert not inf.any()


@pytest.mark.parametrize("curve", CURVES)
def test_ecdsa_vectors_fixed_base_and_verify(oracle, curve):
    """{p256,p384,k256}/src/test_vectors/ecdsa.rs: d*G == Q, x(k*G) mod n == r (signing path,
    mul_by_generator) and x(u1*G + u2*Q) mod n == r (verification path,
    mul_by_generator_and_mul_add_vartime)."""
    c = pyec.CURVES[curve]
    for v in load(curve)["ecdsa"]:
        d, k = bytes.fromhex(v["d"]), bytes.fromhex(v["k"])
        out, inf = oracle.batch_mul_base(c.cid, d + k)
        q = bytes(out[: 2 * c.L])
        assert q == bytes.fromhex(v["q_x"]) + bytes.fromhex(v["q_y"])
        rx = int.from_bytes(bytes(out[2 * c.L: 3 * c.L]), "big")
        r, s = int(v["r"], 16), int(v["s"], 16)
        assert rx % c.n == r
        z = int(v["m"], 16)  # prehash, same width as the field for these vectors
        w = pow(s, -1, c.n)
        u1, u2 = z * w % c.n, r * w % c.n
        pt, pinf = oracle.mul_base_and_mul_add_vartime(c.cid, pyec.enc_scalar(c, u1), pyec.enc_scalar(c, u2), q)
        assert pinf == 0 and int.from_bytes(bytes(pt[: c.L]), "big") % c.n == r


@pytest.mark.parametrize("curve", CURVES)
def test_ecdsa_verify_golden_and_model(oracle, curve):
    """ecref_ecdsa_verify_batch: accepts every signature of the reference's ECDSA vectors
    ({p256,p384,k256}/src/test_vectors/ecdsa.rs, run there through `new_verification_test!`), rejects them once any
    field is disturbed, and agrees with the big-integer model on valid / invalid / out-of-range cases."""
    from gpu_common import ecdsa_cases, ecdsa_pack
    c = pyec.CURVES[curve]
    vec = load(curve)["ecdsa"]
    z = b"".join(bytes.fromhex(v["m"]) for v in vec)
    r = b"".join(bytes.fromhex(v["r"]) for v in vec)
    s = b"".join(bytes.fromhex(v["s"]) for v in vec)
    q = b"".join(bytes.fromhex(v["q_x"]) + bytes.fromhex(v["q_y"]) for v in vec)
    assert oracle.ecdsa_verify(c.cid, z, r, s, q).all()
    for field in range(4):
        bufs = [bytearray(z), bytearray(r), bytearray(s), bytearray(q)]
        for i in range(len(vec)):
            width = len(bufs[field]) // len(vec)
            bufs[field][i * width + width - 1] ^= 1 << (i % 7)
        assert not oracle.ecdsa_verify(c.cid, *[bytes(b) for b in bufs]).any()
    zz, rr, ss, qq, exp = ecdsa_pack(ecdsa_cases(c, 0xEC5A + c.cid))
    assert bytes(oracle.ecdsa_verify(c.cid, zz, rr, ss, qq)) == bytes(exp)
    # NORMALIZE_S (k256/src/ecdsa.rs:104-106): high-S signatures are rejected when the caller asks for it
    got = oracle.ecdsa_verify(c.cid, zz, rr, ss, qq, reject_high_s=True)
    half = (c.n - 1) // 2
    want = [int(e and int.from_bytes(ss[i * c.L:(i + 1) * c.L], "big") <= half) for i, e in enumerate(exp)]
    assert list(got) == want and 0 < sum(want) < int(exp.sum())


@pytest.mark.parametrize("name", ["k256_der", "k256_p1363", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_wycheproof_ecdsa_vectors(oracle, name):
    """The reference's Wycheproof ECDSA blobs (k256/src/ecdsa.rs:263-384 incl. the P1363 file; new_wycheproof_test! at
    p256/src/ecdsa.rs:166-168, p384/src/ecdsa.rs:184-186, p224/src/ecdsa.rs:113-115, p521/src/ecdsa.rs:107-109):
    a signature that does not parse must be a pass = 0 vector, and for every one that parses the oracle's verdict
    (and the big-int model's) equals the pass flag."""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    assert p["total"] > 200 and len(p["expect"]) + len(p["unparsed"]) == p["total"]
    assert not [i for i, ok in p["unparsed"] if ok]
    ok = oracle.ecdsa_verify(c.cid, p["z"], p["r"], p["s"], p["q"], p["reject_high_s"])
    assert bytes(ok) == bytes(p["expect"])
    L = c.L
    for i in range(0, len(p["expect"]), 7):                      # the independent model on a stride (it is slow)
        z, r, s = (int.from_bytes(bytes(p[k][L * i: L * i + L]), "big") for k in ("z", "r", "s"))
        Q = pyec.dec_point(c, bytes(p["q"][2 * L * i: 2 * L * i + 2 * L]), 0)
        assert pyec.ecdsa_verify(c, Q, z, r, s, p["reject_high_s"]) == bool(p["expect"][i])


def test_ecdsa_recover_reference_vectors(oracle):
    """ecref_ecdsa_recover_batch on the reference's own recovery vectors: RECOVERY_TEST_VECTORS (k256/src/ecdsa.rs:190-211,
    SHA-256 digests, recovery ids 0 and 1, keys given SEC1-compressed) and the Ethereum example (:233-261: Keccak-256 digest,
    the key of the stated signing key); a flipped parity bit recovers a different key, a flipped digest bit another one."""
    from gpu_common import recovery_golden, recover_pack
    g = recovery_golden()
    assert len(g) == 3
    z, r, s, recid, exp_xy, _ = recover_pack(g, 32)
    out, ok = oracle.ecdsa_recover(0, z, r, s, recid, reject_high_s=True)
    assert ok.all() and bytes(out) == exp_xy
    out2, ok2 = oracle.ecdsa_recover(0, z, r, s, recid ^ 1, reject_high_s=True)
    assert ok2.all() and all(bytes(out2[64 * i: 64 * i + 64]) != exp_xy[64 * i: 64 * i + 64] for i in range(3))
    zb = bytearray(z); zb[31] ^= 1
    out3, ok3 = oracle.ecdsa_recover(0, bytes(zb), r, s, recid, reject_high_s=True)
    assert ok3[0] and bytes(out3[:64]) != exp_xy[:64] and bytes(out3[64:]) == exp_xy[64:]
    # the recovered keys verify, as `recover_from_prehash` itself checks
    assert oracle.ecdsa_verify(0, z, r, s, out, reject_high_s=True).all()


@pytest.mark.parametrize("curve", [c for c in ALL_CURVES if c != "sm2"])
def test_ecdsa_recover_vs_model(oracle, curve):
    """ecref_ecdsa_recover_batch against the big-integer model: signatures with the recovery id of their nonce point, the
    other ids, disturbed fields, range failures, ids above 3, candidates off the curve, x-reduced candidates (x = r + n);
    with the k256 high-S rule the keys of high-S signatures are withheld."""
    from gpu_common import recover_cases, recover_pack
    c = pyec.CURVES[curve]
    cases = recover_cases(c, 0x4EC0 + c.cid, nvalid=6)
    z, r, s, recid, exp_xy, exp_ok = recover_pack(cases, c.L)
    out, ok = oracle.ecdsa_recover(c.cid, z, r, s, recid)
    assert bytes(ok) == bytes(exp_ok) and bytes(out) == exp_xy
    assert 0 < int(ok.sum()) < len(cases)
    assert any(t[3] & 2 and t[4] is not None for t in cases) == (c.p > c.n)
    out_h, ok_h = oracle.ecdsa_recover(c.cid, z, r, s, recid, reject_high_s=True)
    half = (c.n - 1) // 2
    low = np.array([int.from_bytes(t[2], "big") <= half for t in cases])
    assert bytes(ok_h) == bytes((exp_ok & low).astype(np.uint8)) and 0 < int(ok_h.sum()) < int(ok.sum())
    for i in range(len(cases)):
        want = exp_xy[2 * c.L * i: 2 * c.L * (i + 1)] if ok_h[i] else bytes(2 * c.L)
        assert bytes(out_h[2 * c.L * i: 2 * c.L * (i + 1)]) == want


@pytest.mark.parametrize("name", ["k256_der", "k256_p1363", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_wycheproof_ecdsa_vectors_message_level(oracle, name):
    """The same blobs through ecref_ecdsa_verify_msg_batch — `Verifier::verify(msg, &sig)`: digest (the curve's
    `DigestAlgorithm`: SHA-256 / 384 / 224 / 512), bits2field and verification from the raw messages, grouped by message
    length; every verdict equals the pass flag.  (The harness normalises s for k256 before verifying, k256/src/ecdsa.rs.)"""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    groups = wycheproof_lib.by_message_length(p)
    assert sum(len(g[0]) for g in groups.values()) == len(p["expect"]) and len(groups) >= 3
    for ln, (idx, q, m, sg) in groups.items():
        ok = oracle.ecdsa_verify_msg(c.cid, q, m, ln, sg, p["reject_high_s"])
        assert bytes(ok) == bytes(p["expect"][idx]), (name, ln)


@pytest.mark.parametrize("name", ["k256_der", "p256_der", "p384_der", "p224_der", "p521_der"])
def test_wycheproof_vectors_through_recovery(oracle, name):
    """The Wycheproof blobs once more, through public-key recovery: for every vector whose signature parses, one of the four
    recovery ids gives back the vector's public key exactly when the vector is a valid one (a recovered key always verifies
    the signature it came from, so an invalid vector's key can never come out).  Keys off the curve (a few invalid vectors)
    are never recovered either."""
    import wycheproof_lib
    p = wycheproof_lib.prepare(name)
    c = p["curve"]
    z, r, s, ids = wycheproof_lib.recovery_batch(p)
    keys, ok = oracle.ecdsa_recover(c.cid, z, r, s, ids, p["reject_high_s"])
    assert bytes(wycheproof_lib.recovery_matches(p, keys, ok)) == bytes(p["expect"])
    assert int(p["expect"].sum()) > 100


def test_curve_digests_against_hashlib(oracle):
    """ecref_curve_digest: SHA-256 / 384 / 224 / 512 per curve against hashlib on lengths around every padding boundary; p192,
    sm2 and bign256 have no ECDSA digest."""
    import hashlib
    H = {"k256": "sha256", "p256": "sha256", "p384": "sha384", "p224": "sha224", "p521": "sha512", "bp256": "sha256", "bp384": "sha384",
         "bp256t1": "sha256", "bp384t1": "sha384"}
    rng = random.Random(11)
    for name, h in H.items():
        for n in (0, 1, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 239, 240, 300, 1000):
            m = bytes(rng.randrange(256) for _ in range(n))
            assert oracle.curve_digest(pyec.CURVES[name].cid, m) == hashlib.new(h, m).digest()
    for name in ("p192", "sm2", "bign256"):
        with pytest.raises(Exception):
            oracle.curve_digest(pyec.CURVES[name].cid, b"abc")


def test_sm2dsa_verify_reference_vector_and_model(oracle):
    """ecref_sm2dsa_verify_batch (sm2/src/dsa/verifying.rs:138-171): accepts the reference's own test vector
    (sm2/tests/sm2dsa.rs:16-35, with e = SM3(ZA || M) computed by hashlib), and agrees with the big-int model on signatures
    the model makes and on every way of breaking them (ranges, t = 0, wrong key, off-curve key)."""
    from gpu_common import ecdsa_pack, sm2dsa_cases
    e, r, s, q, exp = ecdsa_pack(sm2dsa_cases(0x5D2A))
    assert exp[0] == 1 and 0 < exp.sum() < len(exp)
    assert bytes(oracle.sm2dsa_verify(e, r, s, q)) == bytes(exp)
    c = pyec.CURVES["sm2"]
    for i in range(len(exp)):
        Q = None
        try:
            Q = pyec.dec_point(c, q[64 * i: 64 * i + 64], 0)
        except AssertionError:
            pass
        got = pyec.sm2dsa_verify(c, Q, *(int.from_bytes(b[32 * i: 32 * i + 32], "big") for b in (e, r, s))) if Q else False
        assert got == bool(exp[i]), i


def test_sm2dsa_verify_messages_reference_vector_sm3_and_model(oracle):
    """ecref_sm3 against OpenSSL's SM3 around the block boundaries; ecref_sm2dsa_verify_msg_batch — `VerifyingKey::new(distid,
    pk)?.verify(msg, sig)`, sm2/src/distid.rs:21-44 + sm2/src/dsa/verifying.rs:126-171 — on the reference's message-level
    vector (sm2/tests/sm2dsa.rs:16-35: it verifies; another identifier or message does not) and on model-made signatures
    under three identifiers (empty, the default, 100 bytes) and three message lengths."""
    import hashlib
    from gpu_common import SM2DSA_KAT as K, sm2dsa_msg_cases, sm2dsa_msg_pack
    rng = random.Random(3)
    for n in (0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 128, 300):
        m = bytes(rng.randrange(256) for _ in range(n))
        assert oracle.sm3(m) == hashlib.new("sm3", m).digest()
    pk, sig, msg = bytes.fromhex(K["public_key"])[1:], bytes.fromhex(K["signature"]), K["message"]
    assert oracle.sm2dsa_verify_msg(K["identity"], pk, msg, len(msg), sig)[0] == 1
    assert oracle.sm2dsa_verify_msg(K["identity"] + b"x", pk, msg, len(msg), sig)[0] == 0
    assert oracle.sm2dsa_verify_msg(K["identity"], pk, b"testinh", len(msg), sig)[0] == 0
    for distid, msg_len in ((b"", 0), (b"1234567812345678", 32), (bytes(range(100)), 77)):
        q, m, sg, exp = sm2dsa_msg_pack(sm2dsa_msg_cases(0x5D30 + msg_len, distid, msg_len, nvalid=4))
        assert bytes(oracle.sm2dsa_verify_msg(distid, q, m, msg_len, sg)) == bytes(exp) and 0 < int(exp.sum()) < len(exp)


def test_schnorr_bip340_vectors(oracle):
    """The BIP340 vectors of k256/src/schnorr.rs (0-3 signing, 4-14 verification incl. every documented failure
    mode, 15-18 variable-length messages) through decompress (lift_x) + Schnorr verification."""
    from gpu_common import schnorr_inputs
    c = pyec.CURVES["k256"]
    vec = load("k256")["schnorr"]
    assert len(vec) == 19

    def pubkey_of(sk):
        out, _ = oracle.batch_mul_base(c.cid, sk)
        return bytes(out[:32])

    e, r, s, pxy, liftable, exp = schnorr_inputs(vec, lambda xs, odd: oracle.batch_decompress(c.cid, xs, odd), pubkey_of)
    # vectors 5 (key not on curve) and 14 (key >= p) fail at the lift, as VerifyingKey::from_bytes does
    assert [v["index"] for v, l in zip(vec, liftable) if not l] == [5, 14]
    pxy = pxy.reshape(-1, 64).copy()
    pxy[liftable == 0] = np.frombuffer(pyec.enc_point(c, pyec.G(c))[0], np.uint8)     # any valid point: verdict stays 0
    got = oracle.schnorr_verify(e, r, s, pxy.reshape(-1)) & liftable
    assert list(got) == list(exp)


def test_schnorr_bip340_vectors_from_wire_bytes(oracle):
    """The same 19 vectors through the all-in-one entry (x-only key, message, 64-byte signature): lift_x, the tagged
    challenge hash and the verification all inside the oracle."""
    c = pyec.CURVES["k256"]
    for v in load("k256")["schnorr"]:
        if "public_key" in v:
            pk = bytes.fromhex(v["public_key"])
        else:
            pk = bytes(oracle.batch_mul_base(c.cid, bytes.fromhex(v["secret_key"]))[0][:32])
        msg = bytes.fromhex(v["message"])
        got = oracle.schnorr_verify_raw(pk, msg, len(msg), bytes.fromhex(v["signature"]))
        assert int(got[0]) == (1 if v["valid"] else 0), v["index"]


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_decompress_vs_model(oracle, curve):
    """DecompressPoint::decompress: generator round trip (p256/tests/affine.rs:12-28 compressed basepoint), random x
    with and without a root, both parities, x >= p."""
    c = pyec.CURVES[curve]
    rng = random.Random(0xDEC0 + c.cid)
    xs, odd, exp = [], [], []
    g = pyec.G(c)
    cand = [g[0], pyec.mul(c, 2, g)[0], c.p, c.p + 1 if c.p + 1 < (1 << (8 * c.L)) else c.p, 0, 1, 2, 3] + \
           [rng.randrange(c.p) for _ in range(120)]
    for x in cand:
        for o in (0, 1):
            xs.append(x.to_bytes(c.L, "big")); odd.append(o)
            rhs = (x ** 3 + c.a * x + c.b) % c.p
            y = pyec.sqrt_mod(rhs, c.p)
            if x >= c.p or y is None:
                exp.append(None)
            else:
                exp.append((x, y if y % 2 == o else (c.p - y) % c.p))
    out, ok = oracle.batch_decompress(c.cid, b"".join(xs), np.array(odd, np.uint8))
    for i, e in enumerate(exp):
        rec = bytes(out[i * 2 * c.L:(i + 1) * 2 * c.L])
        if e is None:
            assert ok[i] == 0 and rec == bytes(2 * c.L)
        else:
            assert ok[i] == 1 and rec == e[0].to_bytes(c.L, "big") + e[1].to_bytes(c.L, "big")
    assert bytes(out[: 2 * c.L]) == pyec.enc_point(c, g if g[1] % 2 == 0 else pyec.neg(c, g))[0]


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_variable_base_vs_node_openssl_ecdh(oracle, curve):
    """The oracle's `P * k` against ECDH shared secrets from Node's crypto / OpenSSL (tests/golden/ecdh_node.json,
    generated by tests/golden/gen_ecdh_node.js): the third opinion SURVEY.md §8c asks for where the reference has no
    known-answer test (variable base with P != G)."""
    c = pyec.CURVES[curve]
    rows = json.load(open(os.path.join(GOLDEN, "ecdh_node.json")))[curve]
    k = b"".join(bytes.fromhex(r["d"]) for r in rows)
    p = b"".join(bytes.fromhex(r["qx"]) + bytes.fromhex(r["qy"]) for r in rows)
    pub, pinf = oracle.batch_mul_base(c.cid, k)                      # OpenSSL's d * G
    assert not pinf.any() and bytes(pub) == b"".join(bytes.fromhex(r["px"]) + bytes.fromhex(r["py"]) for r in rows)
    for vt in (False, True):
        out, inf = oracle.batch_mul(c.cid, k, p, vartime=vt)
        got = np.asarray(out).reshape(len(rows), 2 * c.L)[:, : c.L]
        assert not inf.any() and bytes(got.copy().reshape(-1)) == b"".join(bytes.fromhex(r["z"]) for r in rows)


@pytest.mark.parametrize("curve", ["k256", "p256"])
def test_field_doubling_vectors(oracle, curve):
    """k256 field.rs tests / p256 field.rs:219-245: repeated doubling of 1."""
    c = pyec.CURVES[curve]
    vec = load(curve)["field_dbl"]
    cur = (1).to_bytes(c.L, "big")
    for want in vec:
        assert cur.hex() == want
        cur = oracle.field_op(c.cid, 0, cur, cur)


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_field_ops_vs_bigint(oracle, curve):
    """k256 field.rs:586-597 style differential against integers mod p."""
    c = pyec.CURVES[curve]
    rng = random.Random(0xF1E1D + c.cid)
    edge = [0, 1, 2, c.p - 1, c.p - 2, (c.p - 1) // 2, 2 ** (8 * c.L - 1) % c.p, 0xFFFFFFFF, 2 ** 64 - 1]
    vals = edge + [rng.randrange(c.p) for _ in range(200)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        A, B = a.to_bytes(c.L, "big"), b.to_bytes(c.L, "big")
        assert int.from_bytes(oracle.field_op(c.cid, 0, A, B), "big") == (a + b) % c.p
        assert int.from_bytes(oracle.field_op(c.cid, 1, A, B), "big") == (a - b) % c.p
        assert int.from_bytes(oracle.field_op(c.cid, 2, A, B), "big") == (a * b) % c.p
        assert int.from_bytes(oracle.field_op(c.cid, 3, A), "big") == (a * a) % c.p
        assert int.from_bytes(oracle.field_op(c.cid, 5, A), "big") == (-a) % c.p
        inv = int.from_bytes(oracle.field_op(c.cid, 4, A), "big")
        assert inv == (pow(a, -1, c.p) if a else 0)
    # non-canonical input is rejected (from_bytes range check)
    with pytest.raises(oracle.OracleError):
        oracle.field_op(c.cid, 0, c.p.to_bytes(c.L, "big"), (0).to_bytes(c.L, "big"))


def test_radix16_properties(oracle):
    """primeorder/src/tables/radix16.rs:109-172: digit range, reconstruction, zero, padding."""
    rng = random.Random(16)
    for L, D in ((32, 65), (48, 97), (32, 33)):
        nbytes = (D - 1) // 2
        cases = [0, 1, 7, 8, 15, 16, 2 ** (8 * nbytes) - 1, int("88" * nbytes, 16), int("7f" * nbytes, 16)]
        cases += [rng.getrandbits(8 * nbytes) for _ in range(300)]
        for k in cases:
            d = oracle.radix16(k.to_bytes(L, "big"), D)
            assert len(d) == D
            assert all(-8 <= x < 8 for x in d[:-1]) and 0 <= d[-1] <= 1
            assert sum(int(x) << (4 * i) for i, x in enumerate(d)) == k
        assert not oracle.radix16((0).to_bytes(L, "big"), D).any()


def test_wnaf_form_properties(oracle):
    """wnaf/src/lib.rs:70-150 semantics: odd digits, |d| <= 2^(w-1)-1, w-1 zeros after a non-zero
    digit, exact reconstruction (including the trailing carry digit)."""
    rng = random.Random(5)
    for nbytes, bit_len in ((32, 256), (16, 128), (48, 384)):
        cases = [0, 1, 2 ** bit_len - 1, 2 ** (bit_len - 1), int("f0" * nbytes, 16)]
        cases += [rng.getrandbits(bit_len) for _ in range(300)]
        for k in cases:
            d = oracle.wnaf_form(k.to_bytes(nbytes, "little"), bit_len, 5)
            assert len(d) <= bit_len + 1
            assert sum(int(x) << i for i, x in enumerate(d)) == k
            for i, x in enumerate(d):
                if x != 0:
                    assert x % 2 != 0 and abs(int(x)) <= 15
                    assert not d[i + 1: i + 5].any()


def test_k256_glv_decompose(oracle):
    """k256 mul/glv.rs:149-156 + proof :43-146: r1 + r2*lambda == k (mod n), |r_i| < 2^128 after
    sign folding.  Also cross-checks the rounding against exact rational arithmetic."""
    c = pyec.K256
    rng = random.Random(0x61)
    g1 = 0x3086D221A7D46BCDE86C90E49284EB153DAA8A1471E8CA7FE893209A45DBB031
    g2 = 0xE4437ED6010E88286F547FA90ABFE4C4221208AC9DF506C61571B4AE8AC47F71
    mb1 = 0xE4437ED6010E88286F547FA90ABFE4C3
    mb2 = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE8A280AC50774346DD765CDA83DB1562C
    cases = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2 ** 128, pyec.K256_LAMBDA, 2 ** 255 % c.n]
    cases += [rng.randrange(c.n) for _ in range(500)]
    for k in cases:
        r1b, r2b = oracle.k256_glv_decompose(k.to_bytes(32, "big"))
        r1, r2 = int.from_bytes(r1b, "big"), int.from_bytes(r2b, "big")
        c1 = ((k * g1 + (1 << 383)) >> 384) * mb1 % c.n
        c2 = ((k * g2 + (1 << 383)) >> 384) * mb2 % c.n
        e2 = (c1 + c2) % c.n
        e1 = (k - e2 * pyec.K256_LAMBDA) % c.n
        assert (r1, r2) == (e1, e2)
        assert (r1 + r2 * pyec.K256_LAMBDA) % c.n == k
        assert min(r1, c.n - r1) < 2 ** 128 and min(r2, c.n - r2) < 2 ** 128
    with pytest.raises(oracle.OracleError):
        oracle.k256_glv_decompose(c.n.to_bytes(32, "big"))


def _rand_points(c, rng, n):
    pts = []
    for _ in range(n):
        pts.append(pyec.mul(c, rng.randrange(1, c.n), pyec.G(c)))
    return pts


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_variable_base_vs_bigint(oracle, curve):
    """No reference KAT exists for P != G (SURVEY.md §8c 'Gaps'): differential of both variable-base
    drivers against the independent affine model, including the edge scalars/points of §8d."""
    c = pyec.CURVES[curve]
    rng = random.Random(0xBA5E + c.cid)
    pts = _rand_points(c, rng, 6)
    G = pyec.G(c)
    edge_k = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2 ** 128]
    if curve == "k256":
        edge_k.append(pyec.K256_LAMBDA)
    pairs = [(k, P) for k in edge_k for P in (G, pyec.neg(c, G), pts[0])]
    pairs += [(rng.randrange(c.n), P) for P in pts for _ in range(3)]
    pairs += [(rng.randrange(c.n), pyec.INF), (0, pyec.INF)]
    scal = b"".join(pyec.enc_scalar(c, k) for k, _ in pairs)
    enc = [pyec.enc_point(c, P) for _, P in pairs]
    pxy = b"".join(e[0] for e in enc)
    pinf = np.array([e[1] for e in enc], np.uint8)
    for vt in (False, True):
        out, inf = oracle.batch_mul(c.cid, scal, pxy, pinf, vartime=vt)
        for i, (k, P) in enumerate(pairs):
            got = pyec.dec_point(c, bytes(out[2 * c.L * i: 2 * c.L * (i + 1)]), inf[i])
            assert got == pyec.mul(c, k, P), (curve, vt, i)


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_lincomb_vs_bigint_and_identities(oracle, curve):
    """k256/tests/projective.rs:75-140, p256/tests/projective.rs:83-148: lincomb == sum of
    products, lincomb_vartime == lincomb, chunking does not matter; n == 0 gives the identity."""
    c = pyec.CURVES[curve]
    rng = random.Random(0x11CC + c.cid)
    pts = _rand_points(c, rng, 9) + [pyec.INF, pyec.G(c), pyec.neg(c, pyec.G(c))]
    ks = [rng.randrange(c.n) for _ in pts]
    ks[3] = 0
    ks[-1] = ks[-2]  # k*G + k*(-G) cancels
    pts.append(pts[0]); ks.append(c.n - ks[0])  # P0 term cancels too
    scal = b"".join(pyec.enc_scalar(c, k) for k in ks)
    enc = [pyec.enc_point(c, P) for P in pts]
    pxy = b"".join(e[0] for e in enc)
    pinf = np.array([e[1] for e in enc], np.uint8)
    want = pyec.msm(c, ks, pts)
    for vt in (False, True):
        for chunk in (0, 1, 3, 5):
            out, inf = oracle.msm(c.cid, scal, pxy, pinf, chunk=chunk, vartime=vt)
            assert pyec.dec_point(c, bytes(out), inf) == want
    out, inf = oracle.msm(c.cid, b"", b"", None)
    assert inf == 1 and not out.any()
    # three-term proptest shape
    for _ in range(5):
        sub = rng.sample(range(9), 3)
        s3 = b"".join(pyec.enc_scalar(c, ks[i]) for i in sub)
        p3 = b"".join(enc[i][0] for i in sub)
        out, inf = oracle.msm(c.cid, s3, p3)
        assert pyec.dec_point(c, bytes(out), inf) == pyec.msm(c, [ks[i] for i in sub], [pts[i] for i in sub])


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_mul_by_generator_matches_variable_base(oracle, curve):
    """p256/src/arithmetic/tables.rs:64-80, k256/tests/projective.rs mul_by_generator == G * s,
    scalars drawn like the reference generators (32/48 random bytes -> Scalar::reduce)."""
    c = pyec.CURVES[curve]
    rng = np.random.default_rng(0xEC000000 + c.cid)
    n = 64
    raw = rng.integers(0, 256, n * c.L, dtype=np.uint8)
    raw[: c.L] = 0xFF  # forces the reduction branch
    if c.L == 66:
        raw.reshape(n, c.L)[:, 0] &= 1          # p521: 521-bit values (oracle_lib.scalar_reduce does the same)
    scal = oracle.scalar_reduce(c.cid, raw)
    for i in range(n):
        k = int.from_bytes(bytes(raw[c.L * i: c.L * (i + 1)]), "big")
        assert int.from_bytes(bytes(scal[c.L * i: c.L * (i + 1)]), "big") == (k - c.n if k >= c.n else k)
    gxy = pyec.enc_point(c, pyec.G(c))[0] * n
    a, ai = oracle.batch_mul_base(c.cid, scal)
    b, bi = oracle.batch_mul(c.cid, scal, gxy)
    assert bytes(a) == bytes(b) and bytes(ai) == bytes(bi)
    k0 = int.from_bytes(bytes(scal[: c.L]), "big")
    assert pyec.dec_point(c, bytes(a[: 2 * c.L]), ai[0]) == pyec.mul(c, k0, pyec.G(c))


@pytest.mark.parametrize("curve", ALL_CURVES)
def test_batch_normalize_and_validation(oracle, curve):
    """k256/tests/projective.rs batch_normalize == to_affine; identity handling; decode errors."""
    c = pyec.CURVES[curve]
    rng = random.Random(0xB0 + c.cid)
    pts = _rand_points(c, rng, 5)
    xyz = b""
    want = []
    for i, P in enumerate(pts + [pyec.INF]):
        z = rng.randrange(1, c.p)
        if P is pyec.INF:
            xyz += (0).to_bytes(c.L, "big") + (1).to_bytes(c.L, "big") + (0).to_bytes(c.L, "big")
        else:
            xyz += (P[0] * z % c.p).to_bytes(c.L, "big") + (P[1] * z % c.p).to_bytes(c.L, "big") + z.to_bytes(c.L, "big")
        want.append(P)
    out, inf = oracle.batch_normalize(c.cid, xyz)
    for i, P in enumerate(want):
        assert pyec.dec_point(c, bytes(out[2 * c.L * i: 2 * c.L * (i + 1)]), inf[i]) == P
    good = pyec.enc_point(c, pts[0])[0]
    bad = good[:-1] + bytes([good[-1] ^ 1])
    rc, idx = oracle.validate_points(c.cid, good + bad + good)
    assert rc == -3 and idx == 1
    assert oracle.validate_points(c.cid, good + good)[0] == 0
    with pytest.raises(oracle.OracleError) as e:
        oracle.batch_mul_base(c.cid, c.n.to_bytes(c.L, "big"))
    assert e.value.code == -2


def test_baseline_config0_cpu_plumbing_record():
    """BASELINE.json configs[0] (k256 GENERATOR * random Scalar, batch of 1024, CPU reference path, no GPU) as a named
    artefact: `python bench.py --cpu-plumbing` prints one JSON line whose 1024 results passed the model / group-law check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-plumbing"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["config"]["workload"] == "cpu_k256_1024" and rec["config"]["units"] == 1024 and rec["n_gpus"] == 0
    assert rec["check_vs_model"] is True and rec["value"] > 1000


@pytest.mark.parametrize("curve,low_s", [("p256", False), ("k256", True)])
def test_bench_signature_inputs_are_valid_signatures(oracle, curve, low_s):
    """bench.py's signature workloads hold n DISTINCT tuples per input set: tuple i reuses nonce k_(i mod m) under a key d_i and a
    digest z_i of its own, s_i = k^-1 (z_i + r d_i) computed by `_sign_slice` in worker processes.  Here the same helper on a small
    batch, with the points from the oracle: every tuple must verify (p256) resp. recover to its own key (k256, low-S form with the
    recovery id the helper derives) — the reference's `verify_prehashed` / `recover_from_prehash` semantics."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c = pyec.CURVES[curve]
    L, order = c.L, c.n
    rng = np.random.default_rng(0xEC0051F7 + c.cid)
    n, m = 48, 5
    d = oracle.scalar_reduce(c.cid, rng.integers(0, 256, n * L, dtype=np.uint8))
    z = oracle.scalar_reduce(c.cid, rng.integers(0, 256, n * L, dtype=np.uint8))
    k = oracle.scalar_reduce(c.cid, rng.integers(0, 256, m * L, dtype=np.uint8))
    Q, _ = oracle.batch_mul_base(c.cid, d)
    R, _ = oracle.batch_mul_base(c.cid, k)
    rs, kinvs, odds, xhi = [], [], [], []
    for j in range(m):
        kj = int.from_bytes(bytes(k[j * L:(j + 1) * L]), "big")
        xj = int.from_bytes(bytes(R[j * 2 * L:j * 2 * L + L]), "big")
        rs.append(xj % order); kinvs.append(pow(kj, -1, order)); odds.append(int(R[(j + 1) * 2 * L - 1]) & 1); xhi.append(xj >= order)
    sb, ib = bench._sign_slice((bytes(d), bytes(z), rs, kinvs, odds, xhi, L, order, low_s, 0))
    r = np.frombuffer(b"".join(rs[i % m].to_bytes(L, "big") for i in range(n)), np.uint8)
    s = np.frombuffer(sb, np.uint8)
    if low_s:
        keys, ok = oracle.ecdsa_recover(c.cid, z, r, s, np.frombuffer(ib, np.uint8), True)
        assert bool(np.asarray(ok).all()) and bytes(keys) == bytes(Q)
    else:
        assert bool(np.asarray(oracle.ecdsa_verify(c.cid, z, r, s, Q)).all())
    # a second slice with an offset lands on the right nonces
    sb2, _ = bench._sign_slice((bytes(d[7 * L:]), bytes(z[7 * L:]), rs, kinvs, odds, xhi, L, order, low_s, 7))
    assert sb2 == sb[7 * L:]
