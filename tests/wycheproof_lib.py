"""The reference's Wycheproof ECDSA harness (k256/src/ecdsa.rs:263-384; ecdsa_core::new_wycheproof_test! for the
NIST curves), host side: key padding, strict DER / P1363 signature parsing, message digest and bits2field.  What is
left — verify_prehashed on (z, r, s, Q) — is the part the oracle and ecgpu_ecdsa_verify_batch are checked on.

Vectors: tests/golden/wycheproof.json, extracted from the reference's blobby files by
tests/golden/extract_wycheproof.py."""
import hashlib
import json
import os

import numpy as np

import pyec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wycheproof.json")
# DigestAlgorithm impls: k256/src/ecdsa.rs (Sha256), p256/src/ecdsa.rs:72-73, p384/src/ecdsa.rs:69-70,
# p224/src/ecdsa.rs:69-70, p521/src/ecdsa.rs:69-70
DIGEST = {"k256": hashlib.sha256, "p256": hashlib.sha256, "p384": hashlib.sha384, "p224": hashlib.sha224,
          "p521": hashlib.sha512}
# k256 normalises s before verifying because its verifier rejects high s (k256/src/ecdsa.rs:308-318; NORMALIZE_S)
NORMALIZE_S = {"k256": True, "p256": False, "p384": False, "p224": False, "p521": False}

_sets = None


def load_sets():
    global _sets
    if _sets is None:
        with open(GOLDEN) as f:
            _sets = json.load(f)
    return _sets


def set_names():
    return sorted(load_sets())


def element_from_padded_slice(data, L):
    """k256/src/ecdsa.rs:271-289: left-pad short coordinates, strip leading zero bytes of long ones."""
    if len(data) >= L:
        off = len(data) - L
        if any(data[:off]):
            raise ValueError("EcdsaVerifier: point too large")
        return data[off:]
    return bytes(L - len(data)) + data


class DerError(ValueError):
    pass


def _der_header(buf, pos, want_tag):
    """Strict DER TLV header at buf[pos]: returns (content_start, content_end).  Definite, minimal lengths only."""
    if pos >= len(buf) or buf[pos] != want_tag:
        raise DerError("tag")
    pos += 1
    if pos >= len(buf):
        raise DerError("truncated length")
    b = buf[pos]
    pos += 1
    if b < 0x80:
        ln = b
    else:
        nb = b & 0x7F
        if nb == 0 or nb > 4:                   # indefinite form / absurd
            raise DerError("length form")
        if pos + nb > len(buf):
            raise DerError("truncated length")
        if buf[pos] == 0:
            raise DerError("non-minimal length")
        ln = int.from_bytes(buf[pos:pos + nb], "big")
        pos += nb
        if ln < 0x80:
            raise DerError("non-minimal length")
    if pos + ln > len(buf):
        raise DerError("truncated content")
    return pos, pos + ln


def _der_uint(buf, pos):
    """DER INTEGER holding a non-negative value, canonical (der::asn1::UintRef): returns (value bytes, next pos)."""
    s, e = _der_header(buf, pos, 0x02)
    c = buf[s:e]
    if len(c) == 0:
        raise DerError("empty integer")
    if c[0] & 0x80:
        raise DerError("negative integer")
    if len(c) > 1 and c[0] == 0 and not (c[1] & 0x80):
        raise DerError("non-canonical leading zero")
    return c.lstrip(b"\x00"), e


def parse_der_signature(sig, c):
    """`Signature::<C>::from_der`: SEQUENCE { r INTEGER, s INTEGER }, nothing after it, then from_scalars
    (r, s in [1, n)).  Returns (r, s) as ints or raises DerError."""
    s0, e0 = _der_header(sig, 0, 0x30)
    if e0 != len(sig):
        raise DerError("trailing data")
    rb, p = _der_uint(sig, s0)
    sb, p = _der_uint(sig, p)
    if p != e0:
        raise DerError("extra elements")
    if len(rb) > c.L or len(sb) > c.L:
        raise DerError("integer longer than the field")
    r, s = int.from_bytes(rb, "big"), int.from_bytes(sb, "big")
    if not (0 < r < c.n and 0 < s < c.n):
        raise DerError("scalar out of range")
    return r, s


def parse_p1363_signature(sig, c):
    """`Signature::<C>::from_slice`: exactly 2 L bytes, r and s in [1, n)."""
    if len(sig) != 2 * c.L:
        raise DerError("length")
    r, s = int.from_bytes(sig[:c.L], "big"), int.from_bytes(sig[c.L:], "big")
    if not (0 < r < c.n and 0 < s < c.n):
        raise DerError("scalar out of range")
    return r, s


def bits2field(digest, L):
    """ecdsa::hazmat::bits2field: the leftmost L bytes of the digest, left-padded if shorter."""
    if len(digest) >= L:
        return digest[:L]
    return bytes(L - len(digest)) + digest


def prepare(name):
    """One Wycheproof set -> dict with the verification batch of every vector whose signature parses
    (z, r, s, q as packed uint8 arrays, expected verdicts) and the list of vectors that do not parse
    (all of which the harness requires to be pass = 0)."""
    rec = load_sets()[name]
    c = pyec.CURVES[rec["curve"]]
    L = c.L
    strs = [bytes.fromhex(h) for h in rec["strings"]]
    hash_fn = DIGEST[rec["curve"]]
    z = bytearray()
    r = bytearray()
    s = bytearray()
    q = bytearray()
    expect, unparsed, msgs = [], [], []
    for i, (iwx, iwy, imsg, isig, ok) in enumerate(rec["vectors"]):
        x = element_from_padded_slice(strs[iwx], L)
        y = element_from_padded_slice(strs[iwy], L)
        try:
            if rec["encoding"] == "der":
                ri, si = parse_der_signature(strs[isig], c)
            else:
                ri, si = parse_p1363_signature(strs[isig], c)
        except DerError:
            unparsed.append((i, ok))
            continue
        if NORMALIZE_S[rec["curve"]] and si > c.n // 2:
            si = c.n - si
        z += bits2field(hash_fn(strs[imsg]).digest(), L)
        msgs.append(strs[imsg])
        r += ri.to_bytes(L, "big")
        s += si.to_bytes(L, "big")
        q += x + y
        expect.append(ok)
    u8 = lambda b: np.frombuffer(bytes(b), np.uint8).copy()
    return {"curve": c, "z": u8(z), "r": u8(r), "s": u8(s), "q": u8(q), "expect": np.array(expect, np.uint8),
            "unparsed": unparsed, "reject_high_s": NORMALIZE_S[rec["curve"]], "total": len(rec["vectors"]), "msgs": msgs}


def by_message_length(p):
    """The parsed vectors of prepare() grouped by message length, for the message-level entry points (one length per call):
    {length: (indices, q bytes, messages bytes, r || s bytes)}."""
    L = p["curve"].L
    groups = {}
    for i, m in enumerate(p["msgs"]):
        groups.setdefault(len(m), []).append(i)
    out = {}
    for ln, idx in groups.items():
        q = b"".join(bytes(p["q"][2 * L * i: 2 * L * (i + 1)]) for i in idx)
        sg = b"".join(bytes(p["r"][L * i: L * (i + 1)]) + bytes(p["s"][L * i: L * (i + 1)]) for i in idx)
        out[ln] = (idx, q, b"".join(p["msgs"][i] for i in idx), sg)
    return out


def recovery_batch(p):
    """Every parsed vector of prepare() under all four recovery ids: (z, r, s, ids) with 4 consecutive elements per vector,
    for the rule 'some id recovers the vector's key  <=>  the signature is valid for that key' (a recovered key always
    verifies the signature it was recovered from)."""
    L = p["curve"].L
    n = len(p["expect"])
    rep = lambda a: np.repeat(np.asarray(a, np.uint8).reshape(n, L), 4, axis=0).reshape(-1)
    return rep(p["z"]), rep(p["r"]), rep(p["s"]), np.tile(np.arange(4, dtype=np.uint8), n)


def recovery_matches(p, keys, ok):
    """Per vector: does one of its four ids recover exactly the vector's public key?"""
    L = p["curve"].L
    n = len(p["expect"])
    k = np.asarray(keys, np.uint8).reshape(n, 4, 2 * L)
    q = np.asarray(p["q"], np.uint8).reshape(n, 1, 2 * L)
    return ((k == q).all(axis=2) & (np.asarray(ok).reshape(n, 4) != 0)).any(axis=1).astype(np.uint8)
