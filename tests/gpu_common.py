"""Shared helpers of the -m gpu parity tests."""
import importlib
import json
import os

import numpy as np

import oracle_lib
import pyec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVES = ["k256", "p256", "p384", "p224", "p192", "p521"]          # curves with reference KATs in tests/golden/<curve>.json
ALL_CURVES = CURVES + ["sm2", "bp256", "bp384", "bp256t1", "bp384t1"]               # + SURVEY 8(f) rank 4: checked against the oracle, the big-int model, OpenSSL


def load_golden(curve):
    with open(os.path.join(GOLDEN, curve + ".json")) as f:
        return json.load(f)


def ecgpu_module():
    return importlib.import_module("elliptic-curves_amd")


def rand_scalars(curve_id, n, seed):
    """n*L random bytes mapped through Scalar::reduce, like the reference's proptest generators."""
    L = oracle_lib.FIELD_BYTES[curve_id]
    rng = np.random.default_rng(seed)
    return oracle_lib.scalar_reduce(curve_id, rng.integers(0, 256, n * L, dtype=np.uint8))


def edge_scalars(c):
    ks = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, 2 ** 128, 2 ** (8 * c.L - 1) % c.n, int("80" * c.L, 16) % c.n,
          int("7f" * c.L, 16) % c.n, 0xFFFF, 0x10000, 0x8000]
    if c.name == "k256":
        ks.append(pyec.K256_LAMBDA)
    return ks


def ladder_edge_scalars(c):
    """Scalars that drive the variable-base ladder through its corner cases: accumulator = +-(table operand) at
    the last digit (n - 2d), leading zero digits, digit -8 runs, carries into the top digit."""
    ks = list(range(0, 41)) + [c.n - j for j in range(1, 41)]
    for j in (1, 2, 5, 31, 2 * c.L - 2, 2 * c.L - 1):
        for e in (-9, -8, -1, 0, 1, 7, 8):
            ks.append((16 ** j + e) % c.n)
    for pat in ("88", "08", "80", "f8", "78", "8f", "ff", "f7"):
        ks.append(int(pat * c.L, 16) % c.n)
        ks.append(int(pat * (c.L // 2), 16))
    if c.name == "k256":
        # GLV halves: k = r1 + r2*lambda with one half zero, tiny, or with the digit patterns above
        lam = pyec.K256_LAMBDA
        for r1 in (0, 1, -1, 8, -8, 16, 2 ** 127, -(2 ** 127) + 1, int("88" * 16, 16), int("f8" * 16, 16)):
            for r2 in (0, 1, -1, 9, -16, 2 ** 127 - 1, int("08" * 16, 16), -int("78" * 16, 16)):
                ks.append((r1 + r2 * lam) % c.n)
    return ks


def comb_corner_scalars(c, w):
    """Scalars whose signed w-bit digit strings drive the comb's incomplete (Jacobian) additions towards their
    preconditions (accumulator != +-entry): extreme partial sums, one / two non-zero windows only, carries into the
    top window, 2^bits - n (the one value for which acc - entry = -n would need a carry that cannot happen)."""
    bits = 8 * c.L
    nwin = (bits - 1) // w + 1
    half = 1 << (w - 1)
    ks = [(1 << bits) - c.n, ((1 << bits) - c.n) // 2, (1 << (bits - 1)) - 1, (1 << (bits - 2)), 3 << (bits - 3)]
    pats = [[-half] * (nwin - 1) + [1], [half] * (nwin - 1) + [0], [half - 1] * (nwin - 1) + [1],
            [(-half if j % 2 else half) for j in range(nwin - 1)] + [1], [-1] * (nwin - 1) + [1], [1] * nwin]
    for j in (0, 1, nwin - 2):
        for i in (j + 1, nwin - 1):
            if i <= j or i >= nwin:
                continue
            for dj in (1, -1, half, -half):
                for di in (1, 2):
                    d = [0] * nwin
                    d[j], d[i] = dj, di
                    pats.append(d)
        d = [0] * nwin
        d[j] = half
        pats.append(d)
    for d in pats:
        k = sum(x << (w * j) for j, x in enumerate(d))
        if 0 < k < c.n:
            ks += [k, c.n - k]
    return [k % c.n for k in ks]


def msm_exceptional_terms(c, rng, filler=40):
    """(scalars, points) whose sorted bucket runs contain every way the bucket accumulation's incomplete additions
    can meet +-(their running sum): repeated terms, P, Q, P + Q with one scalar (the third doubles the sum), P, Q,
    -(P + Q) (the sum cancels and more terms follow), P, -P, P, and +k / -k of one point.  rng: random.Random."""
    G = pyec.G(c)
    rp = lambda: pyec.mul(c, rng.randrange(1, c.n), G)
    ks, pts = [], []
    def put(k, P):
        ks.append(k % c.n); pts.append(P)
    k = rng.randrange(1, c.n); P = rp()
    for _ in range(5):
        put(k, P)
    k = rng.randrange(1, c.n); P, Q = rp(), rp()
    put(k, P); put(k, Q); put(k, pyec.add(c, P, Q)); put(k, rp())
    k = rng.randrange(1, c.n); P, Q = rp(), rp()
    put(k, P); put(k, Q); put(k, pyec.neg(c, pyec.add(c, P, Q))); put(k, rp()); put(k, rp())
    k = rng.randrange(1, c.n); P = rp()
    put(k, P); put(k, pyec.neg(c, P)); put(k, P)
    k = rng.randrange(1, c.n); P = rp()
    put(k, P); put(-k, P); put(k, P); put(k, P)
    k = 1; P = rp()                                  # small scalars: upper windows empty, window 0 crowded
    put(1, P); put(1, P); put(2, P); put(c.n - 1, P)
    for _ in range(filler):
        put(rng.randrange(c.n), rp())
    order = list(range(len(ks)))
    rng.shuffle(order)
    return [ks[i] for i in order], [pts[i] for i in order]


def scalars_to_int_sum(scalars, L, n_mod):
    """sum of n big-endian L-byte integers mod n_mod, via 32-bit limb column sums (cheap for 2^24 terms)."""
    a = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, L)
    if L % 4:                                        # p521: 66 bytes -> two leading zero bytes
        a = np.concatenate([np.zeros((a.shape[0], 4 - L % 4), np.uint8), a], axis=1)
        L = a.shape[1]
    a = a.reshape(-1, L // 4, 4)
    words = (a[:, :, 0].astype(np.uint64) << 24) | (a[:, :, 1].astype(np.uint64) << 16) | (a[:, :, 2].astype(np.uint64) << 8) | a[:, :, 3].astype(np.uint64)
    total = 0
    ncol = L // 4
    for j in range(ncol):
        col = int(words[:, j].sum(dtype=np.uint64)) if words.shape[0] < (1 << 31) else sum(int(x) for x in words[:, j])
        total += col << (32 * (ncol - 1 - j))
    return total % n_mod


def ecdsa_cases(c, seed, nvalid=12):
    """(z, r, s, qxy bytes, expected bool) tuples: valid signatures by the big-int model plus every way of breaking
    one that the verification equation distinguishes.  z is an L-byte integer that may exceed n (reduced on use)."""
    import random
    rng = random.Random(seed)
    L = c.L
    cases = []
    G = pyec.G(c)

    def enc(z, r, s, Q, exp):
        qxy = pyec.enc_point(c, Q)[0] if not isinstance(Q, bytes) else Q
        cases.append((z.to_bytes(L, "big"), r.to_bytes(L, "big"), s.to_bytes(L, "big"), qxy, exp))

    for i in range(nvalid):
        d = rng.randrange(1, c.n)
        z = rng.randrange(1 << (8 * L)) if i % 3 else rng.randrange(c.n, 1 << (8 * L))   # also digests >= n
        k = rng.randrange(1, c.n)
        Q = pyec.mul(c, d, G)
        r, s = pyec.ecdsa_sign(c, d, z, k)
        if r == 0 or s == 0:
            continue
        enc(z, r, s, Q, True)
        enc(z, r, c.n - s, Q, True)                                  # the other s: also valid (high-S policy aside)
        enc(z ^ 1, r, s, Q, False)                                   # wrong digest
        enc(z, r, (s + 1) % c.n or 1, Q, False)                      # wrong s
        enc(z, (r + 1) % c.n or 1, s, Q, False)                      # wrong r
        enc(z, r, s, pyec.mul(c, d + 1, G), False)                   # wrong key
        enc(z, r, s, pyec.neg(c, Q), False)
        if i < 4:
            enc(z, 0, s, Q, False)                                   # range failures
            enc(z, r, 0, Q, False)
            enc(z, c.n, s, Q, False)
            enc(z, r, c.n, Q, False)
            if r + c.n < (1 << (8 * L)):
                enc(z, r + c.n, s, Q, False)                         # r + n: same residue, out of range
            bad = bytearray(pyec.enc_point(c, Q)[0]); bad[-1] ^= 1
            enc(z, r, s, bytes(bad), False)                          # off-curve key
            enc(z, r, s, (c.p).to_bytes(L, "big") + pyec.enc_point(c, Q)[0][L:], False)   # coordinate >= p
            enc(z, r, s, bytes(2 * L), False)                        # (0, 0)
    return cases


def ecdsa_pack(cases):
    z = b"".join(t[0] for t in cases)
    r = b"".join(t[1] for t in cases)
    s = b"".join(t[2] for t in cases)
    q = b"".join(t[3] for t in cases)
    exp = np.array([1 if t[4] else 0 for t in cases], np.uint8)
    return z, r, s, q, exp


def recovery_golden():
    """The reference's public-key recovery vectors (k256/src/ecdsa.rs:190-211 and the Ethereum example :233-261) as
    (z, r, s, recid, expected affine key bytes) tuples; the digests are SHA-256 resp. Keccak-256 of the messages."""
    import hashlib
    c = pyec.K256
    out = []
    for v in load_golden("k256")["recovery"]:
        sig = bytes.fromhex(v["sig"])
        if v["hash"] == "sha256":
            z = hashlib.sha256(v["msg_ascii"].encode()).digest()
            Q = pyec.lift_x(c, int(v["pk_sec1"][2:], 16), int(v["pk_sec1"][:2], 16) & 1)
        else:
            z = pyec.keccak256(bytes.fromhex(v["msg_hex"]))
            Q = pyec.mul(c, int(v["secret_key"], 16), pyec.G(c))
        out.append((z, sig[:32], sig[32:], v["recid"], pyec.enc_point(c, Q)[0]))
    return out


def recover_cases(c, seed, nvalid=10):
    """(z, r, s, recid, expected key bytes or None) tuples for public-key recovery: signatures by the big-int model with
    the recovery id of their nonce point, the other three ids, broken fields, range failures, x-reduced candidates
    (x(R) = r + n, built from a curve point with n <= x < p where the curve has one: no nonce is known for those, but any
    (r, s, z) recovers to SOME key), and r values whose candidate x is not on the curve."""
    import random
    rng = random.Random(seed)
    L = c.L
    G = pyec.G(c)
    cases = []

    def enc(z, r, s, recid):
        Q = pyec.ecdsa_recover(c, z, r, s, recid) if r < (1 << (8 * L)) and s < (1 << (8 * L)) else None
        cases.append((z.to_bytes(L, "big"), r.to_bytes(L, "big"), s.to_bytes(L, "big"), recid,
                      pyec.enc_point(c, Q)[0] if Q is not None else None))
        return Q

    for i in range(nvalid):
        d = rng.randrange(1, c.n)
        z = rng.randrange(1 << min(8 * L, c.n.bit_length())) if i % 3 else rng.randrange(c.n, min(2 * c.n, 1 << (8 * L)))
        k = rng.randrange(1, c.n)
        R = pyec.mul(c, k, G)
        r, s = pyec.ecdsa_sign(c, d, z, k)
        if r == 0 or s == 0 or R[0] >= c.n:
            continue
        recid = R[1] & 1
        assert enc(z, r, s, recid) == pyec.mul(c, d, G)
        enc(z, r, c.n - s, recid ^ 1)                            # (r, -s) recovers the same key from -R
        enc(z, r, s, recid ^ 1)                                  # another valid-looking key, not d G
        enc(z, r, s, recid | 2)                                  # x = r + n: almost never below p
        enc(z ^ 1, r, s, recid)
        enc(z, r, (s + 1) % c.n or 1, recid)
        if i < 3:
            enc(z, r, s, 4 + recid)                              # ids above 3 do not parse
            enc(z, r, s, 255)
            enc(z, 0, s, recid)
            enc(z, r, 0, recid)
            enc(z, c.n, s, recid)
            enc(z, r, c.n, recid)
    # candidates whose x is not on the curve
    found = 0
    x = rng.randrange(1, c.n)
    while found < 3:
        x += 1
        if pyec.lift_x(c, x, 0) is None:
            enc(rng.randrange(c.n), x, rng.randrange(1, c.n), found & 1)
            found += 1
    # x-reduced candidates: points with n <= x < p (none on curves with p < n)
    if c.p > c.n:
        found, x = 0, c.n
        while found < 3 and x < c.p and x < c.n + 64:
            for odd in (0, 1):
                if pyec.lift_x(c, x, odd) is not None:
                    Q = enc(rng.randrange(c.n), x - c.n, rng.randrange(1, c.n), 2 | odd) if x > c.n else None
                    found += Q is not None
            x += 1
    return cases


def recover_pack(cases, L):
    z = b"".join(t[0] for t in cases)
    r = b"".join(t[1] for t in cases)
    s = b"".join(t[2] for t in cases)
    recid = np.array([t[3] for t in cases], np.uint8)
    exp_ok = np.array([0 if t[4] is None else 1 for t in cases], np.uint8)
    exp_xy = b"".join(t[4] if t[4] is not None else bytes(2 * L) for t in cases)
    return z, r, s, recid, exp_xy, exp_ok


def bip340_challenge(r32, pk32, msg):
    """int(tagged_hash("BIP0340/challenge", r || pk || m)) as 32 big-endian bytes (k256/src/schnorr.rs tagged_hash)."""
    import hashlib
    tag = hashlib.sha256(b"BIP0340/challenge").digest()
    return hashlib.sha256(tag + tag + r32 + pk32 + msg).digest()


def schnorr_inputs(vectors, decompress, pubkey_of):
    """BIP340 vectors -> (e, r, s, p_xy, liftable, expected).  `decompress(xs, odd)` lifts x-only keys (even y), and
    `pubkey_of(sk32)` gives the x-only public key of a secret key (vectors 15-18 carry only the key pair's secret)."""
    e = r = s = b""
    pks = []
    for v in vectors:
        pk = bytes.fromhex(v["public_key"]) if "public_key" in v else pubkey_of(bytes.fromhex(v["secret_key"]))
        sig = bytes.fromhex(v["signature"])
        msg = bytes.fromhex(v["message"])
        pks.append(pk)
        r += sig[:32]; s += sig[32:]
        e += bip340_challenge(sig[:32], pk, msg)
    pxy, okl = decompress(b"".join(pks), np.zeros(len(pks), np.uint8))
    exp = np.array([1 if v["valid"] else 0 for v in vectors], np.uint8)
    return e, r, s, np.asarray(pxy, np.uint8), np.asarray(okl, np.uint8), exp


# the reference's SM2DSA test vector (sm2/tests/sm2dsa.rs:16-31: OpenSSL-generated signature over b"testing" for the identity
# "example@rustcrypto.org"): public key (SEC1 uncompressed), r || s
SM2DSA_KAT = {
    "public_key": "0408D77AE04C01CC4C1104360DD8AF6B6F7DF334283D7C1A6AFD5652407B87BEE5014E2A57C36C150D16324DC664E31E6432359609C4E79847A5B161C8C7364C8A",
    "identity": b"example@rustcrypto.org", "message": b"testing",
    "signature": "d1dcccedd9fb785e0f67c16b7c52901625c0b69de9bca2144acc7be713cad2fcf7d1eae6e3a157b36c65f672f738ca8b46298bf149a6510072c431b49cd88b1c",
}


def sm2dsa_cases(seed, nvalid=10):
    """(e, r, s, q bytes, expected) tuples for SM2DSA on the prehash: the reference's test vector (e = SM3(ZA || M) computed
    here with hashlib's SM3), signatures made by the big-int model, and every way of breaking one."""
    import hashlib
    import random
    c = pyec.CURVES["sm2"]
    rng = random.Random(seed)
    G = pyec.G(c)
    cases = []

    def enc(e, r, s, Q, exp):
        qxy = Q if isinstance(Q, bytes) else pyec.enc_point(c, Q)[0]
        cases.append((e.to_bytes(32, "big"), r.to_bytes(32, "big"), s.to_bytes(32, "big"), qxy, exp))

    pk = bytes.fromhex(SM2DSA_KAT["public_key"])
    Q = (int.from_bytes(pk[1:33], "big"), int.from_bytes(pk[33:], "big"))
    e = int.from_bytes(hashlib.new("sm3", pyec.sm2_za(c, SM2DSA_KAT["identity"], Q) + SM2DSA_KAT["message"]).digest(), "big")
    sig = bytes.fromhex(SM2DSA_KAT["signature"])
    enc(e, int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big"), Q, True)
    enc(e ^ 1, int.from_bytes(sig[:32], "big"), int.from_bytes(sig[32:], "big"), Q, False)
    for i in range(nvalid):
        d = rng.randrange(1, c.n - 1)
        e = rng.randrange(1 << 256) if i % 3 else rng.randrange(c.n, 1 << 256)            # also digests >= n
        Q = pyec.mul(c, d, G)
        sig = None
        while sig is None:
            sig = pyec.sm2dsa_sign(c, d, e, rng.randrange(1, c.n))
        r, s = sig
        enc(e, r, s, Q, True)
        enc(e ^ 2, r, s, Q, False)
        enc(e, r, (s + 1) % c.n or 1, Q, False)
        enc(e, (r + 1) % c.n or 1, s, Q, False)
        enc(e, r, s, pyec.neg(c, Q), False)
        if i < 3:
            enc(e, 0, s, Q, False)
            enc(e, r, 0, Q, False)
            enc(e, c.n, s, Q, False)
            enc(e, r, c.n, Q, False)
            enc(e, r, c.n - r, Q, False)                                                  # t = r + s = 0 mod n
            bad = bytearray(pyec.enc_point(c, Q)[0]); bad[-1] ^= 1
            enc(e, r, s, bytes(bad), False)
            enc(e, r, s, bytes(64), False)
    return cases


# the reference's bign test vector (bignp256/tests/ecdsa.rs:21-35, from STB 34.101.45 via met-10145-10-01.pdf §6.2): public key
# (x || y, little-endian), message, signature S0 || S1
BIGN_KAT = {
    "public_key": "BD1A5650179D79E03FCEE49D4C2BD5DDF54CE46D0CF11E4FF87BF7A890857FD07AC6A60361E8C8173491686D461B2826190C2EDA5909054A9AB84D2AB9D99A90",
    "message": "B194BAC80A08F53B366D008E58",
    "signature": "19D32B7E01E25BAE4A70EB6BCA42602CCA6A13944451BCC5D4C54CFD8737619C328B8A58FB9C68FD17D569F7D06495FB",
}


def bign_cases(seed, nvalid=8):
    """(h, sig, q bytes, expected) tuples for bign verification on the prehash: the reference's vector (h = belt-hash of its
    message by the model), signatures made by the big-int model, and every way of breaking one."""
    import random
    c = pyec.CURVES["bign256"]
    rng = random.Random(seed)
    G = pyec.G(c)
    le = lambda v, n=32: v.to_bytes(n, "little")
    cases = []
    pk = bytes.fromhex(BIGN_KAT["public_key"])
    h = pyec.belt_hash(bytes.fromhex(BIGN_KAT["message"]))
    sig = bytes.fromhex(BIGN_KAT["signature"])
    cases.append((h, sig, pk, True))
    cases.append((bytes([h[0] ^ 1]) + h[1:], sig, pk, False))
    for i in range(nvalid):
        d = rng.randrange(1, c.n - 1)
        Q = pyec.mul(c, d, G)
        q = le(Q[0]) + le(Q[1])
        hv = rng.randrange(1 << 256) if i % 3 else rng.randrange(c.n, 1 << 256)           # also hashes >= q (Scalar::reduce)
        h = le(hv)
        sig = pyec.bign_sign(c, d, h, rng.randrange(1, c.n))
        s0, s1 = int.from_bytes(sig[:16], "little"), int.from_bytes(sig[16:], "little")
        cases.append((h, sig, q, True))
        cases.append((le(hv ^ 2), sig, q, False))
        cases.append((h, le(s0 ^ (1 << rng.randrange(128)), 16) + sig[16:], q, False))
        cases.append((h, sig[:16] + le((s1 + 1) % c.n or 1), q, False))
        nQ = pyec.neg(c, Q)
        cases.append((h, sig, le(nQ[0]) + le(nQ[1]), False))
        if i < 3:
            cases.append((h, bytes(16) + sig[16:], q, False))                              # S0 = 0 does not parse
            cases.append((h, sig[:16] + bytes(32), q, False))                              # S1 = 0
            cases.append((h, sig[:16] + le(c.n), q, False))                                # S1 = q
            cases.append((h, sig[:16] + le(s1 + c.n) if s1 + c.n < 1 << 256 else sig[:16] + le((1 << 256) - 1), q, False))   # S1 >= q
            bad = bytearray(q); bad[0] ^= 1
            cases.append((h, sig, bytes(bad), False))                                      # key off the curve
            cases.append((h, sig, le(Q[0] + c.p if Q[0] + c.p < 1 << 256 else c.p) + le(Q[1]), False))   # coordinate >= p
            cases.append((h, sig, bytes(64), False))
            # R = ((S1 + H) mod q) G + (S0 + 2^128) Q = O: rejected whatever S0 is
            s1o = (-(s0 + 2 ** 128) * d - hv) % c.n
            if s1o:
                cases.append((h, sig[:16] + le(s1o), q, False))
    return cases


def bign_msg_cases(seed, msg_len, nvalid=6):
    """(q bytes, msg, sig, expected) tuples for bign verification of MESSAGES of one length: the model's signatures over
    belt-hash(msg) and broken ones (another message, another key, disturbed halves)."""
    import random
    c = pyec.CURVES["bign256"]
    rng = random.Random(seed)
    G = pyec.G(c)
    le = lambda v, n=32: v.to_bytes(n, "little")
    cases = []
    if msg_len == 13:
        cases.append((bytes.fromhex(BIGN_KAT["public_key"]), bytes.fromhex(BIGN_KAT["message"]), bytes.fromhex(BIGN_KAT["signature"]), True))
    for i in range(nvalid):
        d = rng.randrange(1, c.n - 1)
        Q = pyec.mul(c, d, G)
        q = le(Q[0]) + le(Q[1])
        msg = bytes(rng.randrange(256) for _ in range(msg_len))
        sig = pyec.bign_sign(c, d, pyec.belt_hash(msg), rng.randrange(1, c.n))
        cases.append((q, msg, sig, True))
        if msg_len:
            other = bytearray(msg); other[rng.randrange(msg_len)] ^= 1 << rng.randrange(8)
            cases.append((q, bytes(other), sig, False))
        Q2 = pyec.mul(c, d + 1, G)
        cases.append((le(Q2[0]) + le(Q2[1]), msg, sig, False))
        flip = bytearray(sig); flip[rng.randrange(48)] ^= 1 << rng.randrange(8)
        cases.append((q, msg, bytes(flip), pyec.bign_verify(c, Q, pyec.belt_hash(msg), bytes(flip))))
    return cases


def sm2dsa_msg_cases(seed, distid, msg_len, nvalid=8):
    """(q bytes, msg, sig bytes, expected) tuples for SM2DSA verification of MESSAGES under one distinguishing identifier:
    signatures made by the big-int model over e = SM3(Z || M) (hashlib's SM3), and the ways of breaking one — another
    message, another key (Z changes), disturbed r / s, range failures, an off-curve key."""
    import hashlib
    import random
    c = pyec.CURVES["sm2"]
    rng = random.Random(seed)
    G = pyec.G(c)
    cases = []
    for i in range(nvalid):
        d = rng.randrange(1, c.n - 1)
        Q = pyec.mul(c, d, G)
        msg = bytes(rng.randrange(256) for _ in range(msg_len))
        e = int.from_bytes(hashlib.new("sm3", pyec.sm2_za(c, distid, Q) + msg).digest(), "big")
        sig = None
        while sig is None:
            sig = pyec.sm2dsa_sign(c, d, e, rng.randrange(1, c.n))
        r, s = sig
        q = pyec.enc_point(c, Q)[0]
        enc = lambda rr, ss: rr.to_bytes(32, "big") + ss.to_bytes(32, "big")
        cases.append((q, msg, enc(r, s), True))
        if msg_len:
            other = bytearray(msg); other[rng.randrange(msg_len)] ^= 1 << rng.randrange(8)
            cases.append((q, bytes(other), enc(r, s), False))
        cases.append((pyec.enc_point(c, pyec.mul(c, d + 1, G))[0], msg, enc(r, s), False))
        cases.append((q, msg, enc(r, (s + 1) % c.n or 1), False))
        cases.append((q, msg, enc((r + 1) % c.n or 1, s), False))
        if i < 2:
            cases.append((q, msg, enc(0, s), False))
            cases.append((q, msg, enc(r, c.n), False))
            bad = bytearray(q); bad[-1] ^= 1
            cases.append((bytes(bad), msg, enc(r, s), False))
    return cases


def sm2dsa_msg_pack(cases):
    return (b"".join(t[0] for t in cases), b"".join(t[1] for t in cases), b"".join(t[2] for t in cases),
            np.array([1 if t[3] else 0 for t in cases], np.uint8))
