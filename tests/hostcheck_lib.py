"""ctypes loader for tests/hostcheck/libhostcheck.so (the kernels' arithmetic compiled for the CPU).
Test infrastructure only; see hostcheck.cpp."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
LIB = os.path.join(HERE, "hostcheck", "libhostcheck.so")
CSRC = os.path.join(os.path.dirname(HERE), "elliptic-curves_amd", "csrc")

_u8p = ctypes.POINTER(ctypes.c_uint8)
_lib = None


def build():
    import fcntl
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("ecgpu_verify.h", "ecgpu_field.h", "ecgpu_params.h", "ecgpu_field_consts.h", "ecgpu_point.h", "ecgpu_recode.h", "ecgpu_varmul.h", "ecgpu_ctmul.h", "ecgpu_msm_chunk.h", "ecgpu_modinv.h", "ecgpu_fixedmul.h", "ecgpu_scalar.h", "ecgpu_sha256.h", "ecgpu_hash.h", "ecgpu_sm3.h", "ecgpu_belt.h")]

    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)
    if fresh():
        return
    # pytest-xdist workers arrive here together: one builds (into a temporary name, renamed when complete), the others wait
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not fresh():
            tmp = LIB + ".tmp.%d" % os.getpid()
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", tmp, SRC])
            os.replace(tmp, LIB)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _a(b):
    if b is None:
        return None
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def _p(a):
    return None if a is None else a.ctypes.data_as(_u8p)


L = {0: 32, 1: 32, 2: 48, 3: 32, 4: 28, 5: 24, 6: 66, 7: 32, 8: 48, 9: 32, 10: 48, 11: 32}


def field_op(curve, op, a, b=None):
    out = np.zeros(L[curve], np.uint8)
    A, B = _a(a), _a(b)
    rc = lib().hc_field_op(curve, op, _p(A), _p(B), _p(out))
    assert rc == 0, rc
    return bytes(out)


def bip340_challenge(r, pk, msg):
    out = np.zeros(32, np.uint8)
    R, P, M = _a(r), _a(pk), _a(msg if len(msg) else b"\0")
    assert lib().hc_bip340_challenge(_p(R), _p(P), _p(M), ctypes.c_size_t(len(msg)), _p(out)) == 0
    return bytes(out)


def scalar_op(curve, op, a, b=None):
    out = np.zeros(L[curve], np.uint8)
    A, B = _a(a), _a(b)
    assert lib().hc_scalar_op(curve, op, _p(A), _p(B), _p(out)) == 0
    return bytes(out)


def field_chain(curve, a, b, steps):
    out = np.zeros(L[curve], np.uint8)
    A, B = _a(a), _a(b)
    assert lib().hc_field_chain(curve, _p(A), _p(B), steps, _p(out)) == 0
    return bytes(out)


def point_op(curve, op, p_xy, p_inf=0, q_xy=None, q_inf=0):
    out = np.zeros(2 * L[curve], np.uint8)
    inf = np.zeros(1, np.uint8)
    P, Q = _a(p_xy), _a(q_xy)
    assert lib().hc_point_op(curve, op, _p(P), int(p_inf), _p(Q), int(q_inf), _p(out), _p(inf)) == 0
    return bytes(out), int(inf[0])


def on_curve(curve, xy):
    X = _a(xy)
    return bool(lib().hc_on_curve(curve, _p(X)))


def batch_mul_base(curve, w, scalars, nthreads=1):
    s = _a(scalars)
    n = s.size // L[curve]
    out = np.zeros(n * 2 * L[curve], np.uint8)
    inf = np.zeros(max(n, 1), np.uint8)
    rc = lib().hc_batch_mul_base(curve, w, _p(s), ctypes.c_size_t(n), ctypes.c_size_t(nthreads), _p(out), _p(inf))
    return rc, out, inf[:n]


def batch_mul(curve, scalars, pxy, pinf=None, nthreads=1):
    s, p, pi = _a(scalars), _a(pxy), _a(pinf)
    n = s.size // L[curve]
    out = np.zeros(n * 2 * L[curve], np.uint8)
    inf = np.zeros(max(n, 1), np.uint8)
    rc = lib().hc_batch_mul(curve, _p(s), _p(p), _p(pi), ctypes.c_size_t(n), ctypes.c_size_t(nthreads), _p(out), _p(inf))
    return rc, out, inf[:n]


def batch_mul_base_ct(curve, scalars):
    """fixed_base_mul_ct (ecgpu_ctmul.h) over CPU-built generator LUTs."""
    s = _a(scalars)
    n = s.size // L[curve]
    out = np.zeros(n * 2 * L[curve], np.uint8)
    inf = np.zeros(max(n, 1), np.uint8)
    rc = lib().hc_batch_mul_base_ct(curve, _p(s), ctypes.c_size_t(n), _p(out), _p(inf))
    return rc, out, inf[:n]


def batch_mul_ct(curve, scalars, pxy, pinf=None):
    """var_base_mul_ct (ecgpu_ctmul.h) with a stack table."""
    s, p, pi = _a(scalars), _a(pxy), _a(pinf)
    n = s.size // L[curve]
    out = np.zeros(n * 2 * L[curve], np.uint8)
    inf = np.zeros(max(n, 1), np.uint8)
    rc = lib().hc_batch_mul_ct(curve, _p(s), _p(p), _p(pi), ctypes.c_size_t(n), _p(out), _p(inf))
    return rc, out, inf[:n]


def msm(curve, c, scalars, pxy, pinf=None, chunk=32, glv=None):
    """The Pippenger pipeline on the CPU.  glv: k256 on the GLV halves (True) or the plain folded scalar (False);
    None = both for k256 (the two must agree; the GLV result is returned), plain for the other curves."""
    s, p, pi = _a(scalars), _a(pxy), _a(pinf)
    n = s.size // L[curve]
    res = None
    for g in ((False, True) if glv is None and curve == 0 else (bool(glv),)):
        out = np.zeros(2 * L[curve], np.uint8)
        inf = np.zeros(1, np.uint8)
        rc = lib().hc_msm(curve, c, ctypes.c_size_t(chunk), int(g), _p(s), _p(p), _p(pi), ctypes.c_size_t(n), _p(out), _p(inf))
        cur = (rc, bytes(out), int(inf[0]))
        if res is not None and cur != res:
            raise AssertionError("plain and GLV Pippenger disagree: %r vs %r" % (res, cur))
        res = cur
    return res


def ecdsa_verify(curve, z, r, s, q, reject_high_s=False):
    Z, R, S, Q = _a(z), _a(r), _a(s), _a(q)
    n = Z.size // L[curve]
    ok = np.zeros(n, np.uint8)
    rc = lib().hc_ecdsa_verify(curve, _p(Z), _p(R), _p(S), _p(Q), ctypes.c_size_t(n), int(bool(reject_high_s)), _p(ok))
    assert rc == 0, rc
    return ok


def sm2dsa_verify(e, r, s, q):
    E, R, S, Q = _a(e), _a(r), _a(s), _a(q)
    n = E.size // 32
    ok = np.zeros(n, np.uint8)
    assert lib().hc_sm2dsa_verify(_p(E), _p(R), _p(S), _p(Q), ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def schnorr_verify(e, r, s, p_xy):
    E, R, S, P = _a(e), _a(r), _a(s), _a(p_xy)
    n = E.size // 32
    ok = np.zeros(n, np.uint8)
    assert lib().hc_schnorr_verify(0, _p(E), ctypes.c_size_t(0), _p(R), _p(S), _p(P), ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def schnorr_verify_raw(pk_x, msgs, msg_len, sigs):
    PK, SG = _a(pk_x), _a(sigs)
    M = _a(msgs) if msg_len else np.zeros(1, np.uint8)
    n = PK.size // 32
    ok = np.zeros(n, np.uint8)
    assert lib().hc_schnorr_verify(1, _p(M), ctypes.c_size_t(msg_len), _p(SG), None, _p(PK), ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def ecdsa_hash_msg(curve, msgs, msg_len):
    """z = bits2field(digest(msg)) per message through the device-side hashing code: n*L bytes, or None for a curve without a digest."""
    M = _a(msgs) if msg_len else np.zeros(1, np.uint8)
    n = (M.size // msg_len) if msg_len else 1
    out = np.zeros(n * L[curve], np.uint8)
    rc = lib().hc_ecdsa_hash_msg(curve, _p(M), ctypes.c_size_t(msg_len), ctypes.c_size_t(n), _p(out))
    return None if rc != 0 else bytes(out)


def belt_hash(msg, cut=0):
    M = _a(msg) if len(msg) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    assert lib().hc_belt_hash(_p(M), ctypes.c_size_t(len(msg)), ctypes.c_size_t(cut), _p(out)) == 0
    return bytes(out)


def bign_verify(h, sigs, q):
    H, SG, Q = _a(h), _a(sigs), _a(q)
    n = H.size // 32
    ok = np.zeros(n, np.uint8)
    assert lib().hc_bign_verify(_p(H), _p(SG), _p(Q), ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def bign_verify_msg(q, msgs, msg_len, sigs):
    Q, SG = _a(q), _a(sigs)
    M = _a(msgs) if msg_len else np.zeros(1, np.uint8)
    n = Q.size // 64
    ok = np.zeros(n, np.uint8)
    assert lib().hc_bign_verify_msg(_p(Q), _p(M), ctypes.c_size_t(msg_len), _p(SG), ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def sm3(msg):
    M = _a(msg) if len(msg) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    assert lib().hc_sm3(_p(M), ctypes.c_size_t(len(msg)), _p(out)) == 0
    return bytes(out)


def sm2dsa_verify_msg(distid, q, msgs, msg_len, sigs):
    D = _a(distid) if len(distid) else np.zeros(1, np.uint8)
    Q, SG = _a(q), _a(sigs)
    M = _a(msgs) if msg_len else np.zeros(1, np.uint8)
    n = Q.size // 64
    ok = np.zeros(n, np.uint8)
    assert lib().hc_sm2dsa_verify_msg(_p(D), ctypes.c_size_t(len(distid)), _p(Q), _p(M), ctypes.c_size_t(msg_len), _p(SG),
                                      ctypes.c_size_t(n), _p(ok)) == 0
    return ok


def ecdsa_recover(curve, z, r, s, recid, reject_high_s=False):
    Z, R, S, I = _a(z), _a(r), _a(s), _a(recid)
    n = I.size
    out = np.zeros(n * 2 * L[curve], np.uint8)
    ok = np.zeros(n, np.uint8)
    rc = lib().hc_ecdsa_recover(curve, _p(Z), _p(R), _p(S), _p(I), ctypes.c_size_t(n), int(bool(reject_high_s)), _p(out), _p(ok))
    assert rc == 0, rc
    return out, ok


def decompress(curve, xs, y_is_odd):
    X, O = _a(xs), _a(y_is_odd)
    n = X.size // L[curve]
    out = np.zeros(n * 2 * L[curve], np.uint8)
    ok = np.zeros(n, np.uint8)
    rc = lib().hc_decompress(curve, _p(X), _p(O), ctypes.c_size_t(n), _p(out), _p(ok))
    assert rc == 0, rc
    return out, ok


def table_rule(curve, w, j, e):
    out = np.zeros(2 * L[curve], np.uint8)
    inf = lib().hc_table_rule(curve, w, j, ctypes.c_uint32(e), _p(out))
    return bytes(out), inf


def radix16(be, nl):
    d = np.zeros(8 * nl + 1, np.int8)
    B = _a(be)
    assert lib().hc_radix16(_p(B), nl, d.ctypes.data_as(ctypes.POINTER(ctypes.c_int8))) == 0
    return d


def signed_windows(be, w):
    d = np.zeros(80, np.int32)
    B = _a(be)
    n = lib().hc_signed_windows(_p(B), w, d.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return d[:n] if n > 0 else None


def k256_glv(k_be):
    r1 = np.zeros(32, np.uint8)
    r2 = np.zeros(32, np.uint8)
    K = _a(k_be)
    flags = lib().hc_k256_glv(_p(K), _p(r1), _p(r2))
    return bytes(r1), bytes(r2), flags
