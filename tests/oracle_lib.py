"""ctypes loader for the CPU oracle (oracle/liboracle_ecref.so) — test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_ecref.so")

K256, P256, P384, SM2, P224, P192, P521, BP256, BP384, BP256T1, BP384T1, BIGN256 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
CURVE_IDS = {"k256": K256, "p256": P256, "p384": P384, "sm2": SM2, "p224": P224, "p192": P192, "p521": P521, "bp256": BP256, "bp384": BP384, "bp256t1": BP256T1, "bp384t1": BP384T1, "bign256": BIGN256}
FIELD_BYTES = {K256: 32, P256: 32, P384: 48, SM2: 32, P224: 28, P192: 24, P521: 66, BP256: 32, BP384: 48, BP256T1: 32, BP384T1: 48, BIGN256: 32}

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i8p = ctypes.POINTER(ctypes.c_int8)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ecref_field_bytes.restype = ctypes.c_size_t
    return _lib


def _buf(a):
    """numpy uint8 array (or None) -> ctypes pointer"""
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def _arr(b, n=None):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy() if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)
    return a


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def batch_mul_base(curve, scalars):
    L = FIELD_BYTES[curve]
    s = _arr(scalars)
    n = s.size // L
    out = np.zeros(n * 2 * L, np.uint8)
    inf = np.zeros(n, np.uint8)
    _chk(lib().ecref_batch_mul_base(curve, _buf(s), ctypes.c_size_t(n), _buf(out), _buf(inf)))
    return out, inf


def _host_threads():
    """threads worth starting: the affinity mask capped by the container's cgroup CPU quota"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def _threaded(n, unit_fn, threads=None):
    """unit_fn(lo, hi) over [0, n) in contiguous slices on all host cores (ctypes releases the GIL inside the C call)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or _host_threads()
    per = max(1, -(-n // (threads * 4)))
    spans = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda sp: unit_fn(*sp), spans))


def batch_mul_base_mt(curve, scalars, threads=None):
    """batch_mul_base on every host core: the whole of a BASELINE-sized batch against the oracle in seconds."""
    L = FIELD_BYTES[curve]
    s = _arr(scalars)
    n = s.size // L
    out = np.zeros(n * 2 * L, np.uint8)
    inf = np.zeros(n, np.uint8)
    fn = lib().ecref_batch_mul_base

    def unit(lo, hi):
        _chk(fn(curve, _buf(s[lo * L: hi * L]), ctypes.c_size_t(hi - lo), _buf(out[lo * 2 * L: hi * 2 * L]), _buf(inf[lo:hi])))

    _threaded(n, unit, threads)
    return out, inf


def batch_mul_mt(curve, scalars, points_xy, threads=None):
    """batch_mul (no identity flags) on every host core."""
    L = FIELD_BYTES[curve]
    s, p = _arr(scalars), _arr(points_xy)
    n = s.size // L
    out = np.zeros(n * 2 * L, np.uint8)
    inf = np.zeros(n, np.uint8)
    fn = lib().ecref_batch_mul

    def unit(lo, hi):
        _chk(fn(curve, _buf(s[lo * L: hi * L]), _buf(p[lo * 2 * L: hi * 2 * L]), None, ctypes.c_size_t(hi - lo),
                _buf(out[lo * 2 * L: hi * 2 * L]), _buf(inf[lo:hi])))

    _threaded(n, unit, threads)
    return out, inf


def batch_mul(curve, scalars, points_xy, points_inf=None, vartime=False):
    L = FIELD_BYTES[curve]
    s, p = _arr(scalars), _arr(points_xy)
    pi = None if points_inf is None else _arr(points_inf)
    n = s.size // L
    out = np.zeros(n * 2 * L, np.uint8)
    inf = np.zeros(n, np.uint8)
    fn = lib().ecref_batch_mul_vartime if vartime else lib().ecref_batch_mul
    _chk(fn(curve, _buf(s), _buf(p), _buf(pi), ctypes.c_size_t(n), _buf(out), _buf(inf)))
    return out, inf


def msm(curve, scalars, points_xy, points_inf=None, chunk=0, vartime=False):
    L = FIELD_BYTES[curve]
    s, p = _arr(scalars), _arr(points_xy)
    pi = None if points_inf is None else _arr(points_inf)
    n = s.size // L
    out = np.zeros(2 * L, np.uint8)
    inf = np.zeros(1, np.uint8)
    _chk(lib().ecref_msm(curve, _buf(s), _buf(p), _buf(pi), ctypes.c_size_t(n), ctypes.c_size_t(chunk),
                         int(vartime), _buf(out), _buf(inf)))
    return out, int(inf[0])


def mul_base_and_mul_add_vartime(curve, a, b, p_xy, p_inf=0):
    L = FIELD_BYTES[curve]
    out = np.zeros(2 * L, np.uint8)
    inf = np.zeros(1, np.uint8)
    _chk(lib().ecref_mul_base_and_mul_add_vartime(curve, _buf(_arr(a)), _buf(_arr(b)), _buf(_arr(p_xy)), int(p_inf),
                                                  _buf(out), _buf(inf)))
    return out, int(inf[0])


def ecdsa_verify(curve, z, r, s, q_xy, reject_high_s=False):
    L = FIELD_BYTES[curve]
    zz, rr, ss, qq = _arr(z), _arr(r), _arr(s), _arr(q_xy)
    n = zz.size // L
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_ecdsa_verify_batch(curve, _buf(zz), _buf(rr), _buf(ss), _buf(qq), ctypes.c_size_t(n),
                                        int(bool(reject_high_s)), _buf(ok)))
    return ok


def ecdsa_recover(curve, z, r, s, recid, reject_high_s=False):
    L = FIELD_BYTES[curve]
    zz, rr, ss, ii = _arr(z), _arr(r), _arr(s), _arr(recid)
    n = ii.size
    out = np.zeros(n * 2 * L, np.uint8)
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_ecdsa_recover_batch(curve, _buf(zz), _buf(rr), _buf(ss), _buf(ii), ctypes.c_size_t(n),
                                         int(bool(reject_high_s)), _buf(out), _buf(ok)))
    return out, ok


def schnorr_verify(e, r, s, p_xy):
    ee, rr, ss, pp = _arr(e), _arr(r), _arr(s), _arr(p_xy)
    n = ee.size // 32
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_schnorr_verify_batch(_buf(ee), _buf(rr), _buf(ss), _buf(pp), ctypes.c_size_t(n), _buf(ok)))
    return ok


def sm2dsa_verify(e, r, s, q_xy):
    ee, rr, ss, qq = _arr(e), _arr(r), _arr(s), _arr(q_xy)
    n = ee.size // 32
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_sm2dsa_verify_batch(_buf(ee), _buf(rr), _buf(ss), _buf(qq), ctypes.c_size_t(n), _buf(ok)))
    return ok


def curve_digest(curve, msg):
    m = _arr(msg) if len(msg) else None
    out = np.zeros(64, np.uint8)
    ln = ctypes.c_size_t(0)
    _chk(lib().ecref_curve_digest(curve, _buf(m), ctypes.c_size_t(len(msg)), _buf(out), ctypes.byref(ln)))
    return bytes(out[: ln.value])


def ecdsa_verify_msg(curve, q_xy, msgs, msg_len, sigs, reject_high_s=False):
    L = FIELD_BYTES[curve]
    qq, sg = _arr(q_xy), _arr(sigs)
    mm = _arr(msgs) if msg_len else None
    n = qq.size // (2 * L)
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_ecdsa_verify_msg_batch(curve, _buf(qq), _buf(mm), ctypes.c_size_t(msg_len), _buf(sg), ctypes.c_size_t(n),
                                            int(bool(reject_high_s)), _buf(ok)))
    return ok


def sm3(msg):
    m = _arr(msg) if len(msg) else None
    out = np.zeros(32, np.uint8)
    _chk(lib().ecref_sm3(_buf(m), ctypes.c_size_t(len(msg)), _buf(out)))
    return bytes(out)


def sm2dsa_verify_msg(distid, q_xy, msgs, msg_len, sigs):
    qq, sg = _arr(q_xy), _arr(sigs)
    dd = _arr(distid) if len(distid) else None
    mm = _arr(msgs) if msg_len else None
    n = qq.size // 64
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_sm2dsa_verify_msg_batch(_buf(dd), ctypes.c_size_t(len(distid)), _buf(qq), _buf(mm), ctypes.c_size_t(msg_len),
                                             _buf(sg), ctypes.c_size_t(n), _buf(ok)))
    return ok


def belt_hash(msg):
    m = _arr(msg) if len(msg) else None
    out = np.zeros(32, np.uint8)
    lib().ecref_belt_hash(_buf(m), ctypes.c_size_t(len(msg)), _buf(out))
    return bytes(out)


def bign_verify(h, sigs, q_xy):
    hh, sg, qq = _arr(h), _arr(sigs), _arr(q_xy)
    n = hh.size // 32
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_bign_verify_batch(_buf(hh), _buf(sg), _buf(qq), ctypes.c_size_t(n), _buf(ok)))
    return ok


def bign_verify_msg(q_xy, msgs, msg_len, sigs):
    qq, sg = _arr(q_xy), _arr(sigs)
    mm = _arr(msgs) if msg_len else None
    n = qq.size // 64
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_bign_verify_msg_batch(_buf(qq), _buf(mm), ctypes.c_size_t(msg_len), _buf(sg), ctypes.c_size_t(n), _buf(ok)))
    return ok


def schnorr_verify_raw(pk_x, msgs, msg_len, sigs):
    pk, sg = _arr(pk_x), _arr(sigs)
    mm = _arr(msgs) if msg_len else None
    n = pk.size // 32
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_schnorr_verify_raw_batch(_buf(pk), _buf(mm), ctypes.c_size_t(msg_len), _buf(sg), ctypes.c_size_t(n), _buf(ok)))
    return ok


def batch_decompress(curve, xs, y_is_odd):
    L = FIELD_BYTES[curve]
    x, odd = _arr(xs), _arr(y_is_odd)
    n = x.size // L
    out = np.zeros(n * 2 * L, np.uint8)
    ok = np.zeros(n, np.uint8)
    _chk(lib().ecref_batch_decompress(curve, _buf(x), _buf(odd), ctypes.c_size_t(n), _buf(out), _buf(ok)))
    return out, ok


def field_op(curve, op, a, b=None):
    L = FIELD_BYTES[curve]
    out = np.zeros(L, np.uint8)
    _chk(lib().ecref_field_op(curve, op, _buf(_arr(a)), None if b is None else _buf(_arr(b)), _buf(out)))
    return bytes(out)


def point_op(curve, op, p_xy, p_inf=0, q_xy=None, q_inf=0):
    L = FIELD_BYTES[curve]
    out = np.zeros(2 * L, np.uint8)
    inf = np.zeros(1, np.uint8)
    _chk(lib().ecref_point_op(curve, op, _buf(_arr(p_xy)), int(p_inf), None if q_xy is None else _buf(_arr(q_xy)),
                              int(q_inf), _buf(out), _buf(inf)))
    return bytes(out), int(inf[0])


def batch_normalize(curve, xyz):
    L = FIELD_BYTES[curve]
    a = _arr(xyz)
    n = a.size // (3 * L)
    out = np.zeros(n * 2 * L, np.uint8)
    inf = np.zeros(n, np.uint8)
    _chk(lib().ecref_batch_normalize(curve, _buf(a), ctypes.c_size_t(n), _buf(out), _buf(inf)))
    return out, inf


def radix16(scalar_be, ndigits):
    s = _arr(scalar_be)
    d = np.zeros(ndigits, np.int8)
    _chk(lib().ecref_radix16(_buf(s), ctypes.c_size_t(s.size), ndigits, d.ctypes.data_as(_i8p)))
    return d


def wnaf_form(le_bytes, bit_len, window=5):
    s = _arr(le_bytes)
    d = np.zeros(bit_len + 1, np.int8)
    n = lib().ecref_wnaf_form(_buf(s), ctypes.c_size_t(s.size), ctypes.c_size_t(bit_len), window,
                              d.ctypes.data_as(_i8p))
    return d[:n]


def k256_glv_decompose(k_be):
    r1 = np.zeros(32, np.uint8)
    r2 = np.zeros(32, np.uint8)
    _chk(lib().ecref_k256_glv_decompose(_buf(_arr(k_be)), _buf(r1), _buf(r2)))
    return bytes(r1), bytes(r2)


def validate_points(curve, points_xy, points_inf=None):
    L = FIELD_BYTES[curve]
    p = _arr(points_xy)
    pi = None if points_inf is None else _arr(points_inf)
    bad = ctypes.c_size_t(0)
    rc = lib().ecref_validate_points(curve, _buf(p), _buf(pi), ctypes.c_size_t(p.size // (2 * L)), ctypes.byref(bad))
    return rc, bad.value


def scalar_reduce(curve, scalars):
    """Scalar::reduce(bytes) of the reference's proptest generators, applied to a uint8 array."""
    L = FIELD_BYTES[curve]
    s = _arr(scalars).copy()
    if curve == P521:          # 66 bytes hold 528 bits but n < 2^521: keep 521 so that one subtraction of n reduces
        s.reshape(-1, L)[:, 0] &= 1
    _chk(lib().ecref_scalar_reduce(curve, _buf(s), ctypes.c_size_t(s.size // L)))
    return s
