"""ecgpu — Python (ctypes) binding of libecgpu.so, the MI355X batch scalar-multiplication / MSM engine.

This is plumbing around the C ABI of include/ecgpu.h (bench.py, the tests and torch-based callers use
it); the product is the shared library.  The directory name contains a hyphen, so import it with

    import importlib; ecgpu = importlib.import_module("elliptic-curves_amd")

There is no CPU fallback: `Engine()` raises `EcgpuError` when the HIP extension is missing or no
gfx950 device is usable.

In a process that also uses torch, import torch FIRST: libecgpu.so then binds to the HIP runtime torch ships.  The
other order leaves two HIP runtimes in the process and torch reports "No HIP GPUs are available"
(tests/gpu_dev_pointer_check.py is the working arrangement).

Operation names follow the reference's trait surface (RustCrypto/elliptic-curves):
    mul_by_generator          ProjectivePoint::mul_by_generator / MulBackend::mul_by_generator
    mul                       impl Mul<Scalar> for ProjectivePoint
    lincomb                   LinearCombination::lincomb
    mul_by_generator_and_mul_add   MulByGeneratorVartime::mul_by_generator_and_mul_add_vartime
    batch_normalize           BatchNormalize::batch_normalize
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libecgpu.so")

K256, P256, P384, SM2, P224, P192, P521, BP256, BP384, BP256T1, BP384T1, BIGN256 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
CURVE_IDS = {"k256": K256, "p256": P256, "p384": P384, "sm2": SM2, "p224": P224, "p192": P192, "p521": P521, "bp256": BP256, "bp384": BP384, "bp256t1": BP256T1, "bp384t1": BP384T1, "bign256": BIGN256}
FIELD_BYTES = {K256: 32, P256: 32, P384: 48, SM2: 32, P224: 28, P192: 24, P521: 66, BP256: 32, BP384: 48, BP256T1: 32, BP384T1: 48, BIGN256: 32}

OK = 0
ERR_CURVE, ERR_SCALAR_RANGE, ERR_POINT, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_ARG = -1, -2, -3, -4, -5, -6, -7

# every symbol include/ecgpu.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "ecgpu_init", "ecgpu_device_count", "ecgpu_destroy", "ecgpu_last_error", "ecgpu_field_bytes", "ecgpu_set_stream",
    "ecgpu_set_base_window", "ecgpu_set_msm_window", "ecgpu_batch_mul_base", "ecgpu_batch_mul", "ecgpu_msm",
    "ecgpu_batch_mul_base_and_mul_add", "ecgpu_batch_normalize", "ecgpu_batch_mul_base_dev",
    "ecgpu_batch_mul_dev", "ecgpu_msm_dev", "ecgpu_batch_mul_base_and_mul_add_dev", "ecgpu_batch_normalize_dev",
    "ecgpu_point_sum", "ecgpu_point_sum_dev", "ecgpu_k256_glv_decompose", "ecgpu_valu_probe",
    "ecgpu_last_timing", "ecgpu_set_timing", "ecgpu_version", "ecgpu_ecdsa_verify_batch", "ecgpu_ecdsa_verify_batch_dev",
    "ecgpu_schnorr_verify_batch", "ecgpu_schnorr_verify_batch_dev", "ecgpu_batch_decompress",
    "ecgpu_batch_decompress_dev", "ecgpu_batch_ecdh", "ecgpu_batch_ecdh_dev",
    "ecgpu_schnorr_verify_raw_batch", "ecgpu_schnorr_verify_raw_batch_dev", "ecgpu_host_alloc", "ecgpu_host_free",
    "ecgpu_batch_mul_base_compressed", "ecgpu_batch_mul_base_compressed_dev",
    "ecgpu_dev_alloc", "ecgpu_dev_free", "ecgpu_copy_to_device", "ecgpu_copy_to_host",
    "ecgpu_msm_parts_bytes", "ecgpu_msm_plan_window", "ecgpu_msm_parts_dev", "ecgpu_msm_finish_dev",
    "ecgpu_group_init", "ecgpu_group_destroy", "ecgpu_group_size", "ecgpu_group_ctx", "ecgpu_group_last_error",
    "ecgpu_group_exchange", "ecgpu_group_set_msm_window", "ecgpu_group_msm", "ecgpu_group_msm_dev",
    "ecgpu_group_batch_mul_base", "ecgpu_group_batch_mul", "ecgpu_selftest_field", "ecgpu_selftest_point",
    "ecgpu_sm2dsa_verify_batch", "ecgpu_sm2dsa_verify_batch_dev", "ecgpu_set_async", "ecgpu_synchronize",
    "ecgpu_ecdsa_recover_batch", "ecgpu_ecdsa_recover_batch_dev",
    "ecgpu_sm2dsa_verify_msg_batch", "ecgpu_sm2dsa_verify_msg_batch_dev",
    "ecgpu_set_msm_lanes", "ecgpu_bign_verify_batch", "ecgpu_bign_verify_batch_dev", "ecgpu_bign_verify_msg_batch", "ecgpu_bign_verify_msg_batch_dev",
    "ecgpu_ecdsa_verify_msg_batch", "ecgpu_ecdsa_verify_msg_batch_dev",
    "ecgpu_group_ecdsa_verify_batch", "ecgpu_group_ecdsa_verify_msg_batch", "ecgpu_group_ecdsa_recover_batch",
    "ecgpu_batch_mul_base_ct", "ecgpu_batch_mul_base_ct_dev", "ecgpu_batch_mul_ct", "ecgpu_batch_mul_ct_dev",
    "ecgpu_batch_ecdh_ct", "ecgpu_batch_ecdh_ct_dev",
    "ecgpu_lincomb_ct", "ecgpu_lincomb_ct_dev", "ecgpu_msm_compressed", "ecgpu_msm_compressed_dev",
    "ecgpu_batch_mul_compressed", "ecgpu_batch_mul_compressed_dev", "ecgpu_wipe", "ecgpu_group_exchange_reason",
    "ecgpu_set_table_policy", "ecgpu_set_table_budget", "ecgpu_base_table_info", "ecgpu_group_set_exchange_timeout", "ecgpu_group_set_exchange", "ecgpu_msm_parts_join_dev",
]
TABLE_ADAPTIVE, TABLE_EAGER = 0, 1
EXCHANGE_PEER, EXCHANGE_RCCL = 1, 2


GROUP_ORDERS = {   # k256/src/lib.rs:71, p256/src/lib.rs:60, p384/src/lib.rs:73
    0: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    1: 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    2: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973,
    3: 0xFFFFFFFEFFFFFFFFFFFFFFFFFFFFFFFF7203DF6B21C6052B53BBF40939D54123,   # sm2/src/lib.rs:86
    4: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFF16A2E0B8F03E13DD29455C5C2A3D,           # p224/src/lib.rs:50-55
    5: 0xFFFFFFFFFFFFFFFFFFFFFFFF99DEF836146BC9B1B4D22831,                   # p192/src/lib.rs:41
    6: 0x01FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFA51868783BF2F966B7FCC0148F709A5D03BB5C9B8899C47AEBB6FB71E91386409,   # p521/src/lib.rs:51-60
    7: 0xA9FB57DBA1EEA9BC3E660A909D838D718C397AA3B561A6F7901E0E82974856A7,   # bp256/src/lib.rs:70
    8: 0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B31F166E6CAC0425A7CF3AB6AF6B7FC3103B883202E9046565,   # bp384/src/lib.rs:73
    9: 0xA9FB57DBA1EEA9BC3E660A909D838D718C397AA3B561A6F7901E0E82974856A7,   # bp256/src/t1.rs:36 (the r1 order)
    10: 0x8CB91E82A3386D280F5D6F7E50E641DF152F7109ED5456B31F166E6CAC0425A7CF3AB6AF6B7FC3103B883202E9046565,   # bp384/src/t1.rs (the r1 order)
    11: 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFD95C8ED60DFB4DFC7E5ABF99263D6607,   # bignp256/src/lib.rs:74
}


class EcgpuError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("ecgpu error %d%s" % (code, (": " + msg) if msg else ""))
        self.code = code


_u8p = ctypes.POINTER(ctypes.c_uint8)
_libs = {}


def load_library(variant=None):
    """Loads libecgpu.so (built in-tree by `make -C elliptic-curves_amd` / __graft_entry__.build()).

    `variant` names another build of this tree that lives beside it, lib/libecgpu_<variant>.so — "knobs" is the tool build in
    which the ECGPU_* tuning knobs are read from the environment (csrc/ecgpu_knobs.h; the product never reads it); the A/B
    recipes of tools/gpu_run.sh name theirs through ECGPU_TOOL_LIB, which replaces the DEFAULT library of the process.
    Either way only files lib/libecgpu_<suffix>.so of this tree can be selected: the loader cannot be pointed anywhere else."""
    if variant in _libs:
        return _libs[variant]
    path = LIB_PATH
    alt = os.path.join(os.path.dirname(LIB_PATH), "libecgpu_%s.so" % variant) if variant else os.environ.get("ECGPU_TOOL_LIB")
    if alt:
        alt = os.path.realpath(alt)
        if os.path.dirname(alt) != os.path.realpath(os.path.dirname(LIB_PATH)) or not os.path.basename(alt).startswith("libecgpu_"):
            raise EcgpuError(ERR_ARG, "ECGPU_TOOL_LIB must name a lib/libecgpu_<suffix>.so of this tree")
        path = alt
    if not os.path.exists(path):
        raise EcgpuError(ERR_NO_DEVICE, "HIP extension %s is missing; run __graft_entry__.build()" % path)
    lib = ctypes.CDLL(path)
    lib.ecgpu_last_error.restype = ctypes.c_char_p
    lib.ecgpu_version.restype = ctypes.c_char_p
    lib.ecgpu_field_bytes.restype = ctypes.c_size_t
    lib.ecgpu_host_alloc.restype = ctypes.c_void_p
    lib.ecgpu_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.ecgpu_host_free.restype = None
    lib.ecgpu_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.ecgpu_dev_alloc.restype = ctypes.c_void_p
    lib.ecgpu_dev_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.ecgpu_dev_free.restype = None
    lib.ecgpu_dev_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.ecgpu_copy_to_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.ecgpu_copy_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.ecgpu_group_last_error.restype = ctypes.c_char_p
    lib.ecgpu_group_exchange.restype = ctypes.c_char_p
    lib.ecgpu_group_exchange_reason.restype = ctypes.c_char_p
    lib.ecgpu_group_exchange_reason.argtypes = [ctypes.c_void_p]
    lib.ecgpu_group_last_error.argtypes = [ctypes.c_void_p]
    lib.ecgpu_group_exchange.argtypes = [ctypes.c_void_p]
    lib.ecgpu_group_destroy.restype = None
    lib.ecgpu_group_destroy.argtypes = [ctypes.c_void_p]
    lib.ecgpu_msm_parts_bytes.restype = ctypes.c_size_t
    lib.ecgpu_msm_parts_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    _libs[variant] = lib
    return lib


def _field_bytes(curve):
    if curve not in FIELD_BYTES:
        raise EcgpuError(ERR_CURVE, "unknown curve id %r" % (curve,))
    return FIELD_BYTES[curve]


def _host(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return np.ascontiguousarray(a, dtype=np.uint8)
    return np.frombuffer(bytes(a), dtype=np.uint8).copy()


def _hp(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _need(what, a, nbytes):
    """Host buffers are handed to C as bare pointers: a short one would be read or written past its end."""
    if a is not None and a.size != nbytes:
        raise EcgpuError(ERR_ARG, "%s holds %d bytes, expected %d" % (what, a.size, nbytes))


def _dp(t):
    """device pointer of a torch tensor / DeviceBuffer / int / None"""
    if t is None:
        return None
    if isinstance(t, int):
        return ctypes.c_void_p(t)
    return ctypes.c_void_p(t.data_ptr())


class DeviceBuffer:
    """Device memory from ecgpu_dev_alloc; `buf.at(offset)` is a raw device address usable as any *_dev argument."""

    def __init__(self, eng, ptr, nbytes):
        self._eng, self.ptr, self.nbytes = eng, ptr, nbytes

    def data_ptr(self):
        return self.ptr

    def at(self, offset):
        return self.ptr + offset

    def free(self):
        if self.ptr and getattr(self._eng, "_ctx", None):
            self._eng._lib.ecgpu_dev_free(self._eng._ctx, ctypes.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One context = one GPU, one stream, device-resident basepoint tables.

    The *_dev methods enqueue on the context's own non-blocking stream: tensors produced on torch's current stream are
    not ordered before them unless that stream is handed over first,
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    (bench.py does) or synchronised."""

    def __init__(self, device=0, variant=None):
        self._lib = load_library(variant)       # variant="knobs": the tool build that reads the ECGPU_* tuning knobs (load_library)
        self._pinned = {}
        self._ctx = ctypes.c_void_p()
        rc = self._lib.ecgpu_init(ctypes.byref(self._ctx), int(device))
        if rc != OK:
            self._ctx = None
            raise EcgpuError(rc, "ecgpu_init failed (no gfx950 device?)")
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.ecgpu_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise EcgpuError(rc, (self._lib.ecgpu_last_error(self._ctx) or b"").decode())

    # ---- configuration ----
    def set_stream(self, stream_ptr):
        self._chk(self._lib.ecgpu_set_stream(self._ctx, ctypes.c_void_p(stream_ptr or 0)))

    def host_alloc(self, nbytes):
        """uint8 numpy array over page-locked host memory (ecgpu_host_alloc): buffers of the host-pointer calls that live
        here are transferred at PCIe DMA speed instead of through the driver's bounce buffers.  Free with host_free."""
        p = self._lib.ecgpu_host_alloc(self._ctx, ctypes.c_size_t(nbytes))
        if not p:
            raise EcgpuError(-1, "ecgpu_host_alloc(%d) failed" % nbytes)
        arr = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p))
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p:
            self._lib.ecgpu_host_free(self._ctx, ctypes.c_void_p(p))

    # ---- device memory without torch (ecgpu_dev_alloc & co): DeviceBuffer objects work wherever a tensor does ----
    def dev_alloc(self, nbytes):
        p = self._lib.ecgpu_dev_alloc(self._ctx, ctypes.c_size_t(nbytes))
        if not p:
            raise EcgpuError(ERR_OOM, "ecgpu_dev_alloc(%d) failed" % nbytes)
        return DeviceBuffer(self, p, nbytes)

    def to_device(self, host, buf=None):
        """numpy uint8 / bytes -> DeviceBuffer (a new one unless `buf` is given)."""
        h = _host(host)
        buf = buf if buf is not None else self.dev_alloc(max(h.size, 16))
        if h.size > buf.nbytes:
            raise EcgpuError(ERR_ARG, "device buffer too small")
        self._chk(self._lib.ecgpu_copy_to_device(self._ctx, ctypes.c_void_p(buf.ptr), h.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_size_t(h.size)))
        return buf

    def to_host(self, buf, nbytes=None, offset=0):
        nbytes = buf.nbytes - offset if nbytes is None else nbytes
        if offset + nbytes > buf.nbytes:
            raise EcgpuError(ERR_ARG, "read past the end of the device buffer")
        out = np.empty(nbytes, np.uint8)
        self._chk(self._lib.ecgpu_copy_to_host(self._ctx, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(buf.ptr + offset),
                                               ctypes.c_size_t(nbytes)))
        return out

    def set_base_window(self, curve, bits):
        """Pins the comb width of `curve` (4..26); 0 returns it to the table policy."""
        self._chk(self._lib.ecgpu_set_base_window(self._ctx, curve, bits))

    def set_table_policy(self, policy):
        """TABLE_ADAPTIVE (default: the generator table grows with the work the device has seen) or TABLE_EAGER (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_set_table_policy(self._ctx, int(policy)))

    def set_table_budget(self, max_table_bytes):
        self._chk(self._lib.ecgpu_set_table_budget(self._ctx, ctypes.c_size_t(int(max_table_bytes))))

    def base_table_info(self, curve):
        """-> {"window_bits", "bytes", "build_ms"} of the comb table `curve` uses on this context now (window_bits 0: none yet)."""
        w, b, ms = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_double(0)
        self._chk(self._lib.ecgpu_base_table_info(self._ctx, curve, ctypes.byref(w), ctypes.byref(b), ctypes.byref(ms)))
        return {"window_bits": w.value, "bytes": b.value, "build_ms": ms.value}

    def set_msm_window(self, bits):
        self._chk(self._lib.ecgpu_set_msm_window(self._ctx, bits))

    def set_async(self, on=True):
        """Device-pointer calls return once queued; input-check errors surface at synchronize() (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_set_async(self._ctx, 1 if on else 0))

    def synchronize(self):
        self._chk(self._lib.ecgpu_synchronize(self._ctx))

    def set_timing(self, on=True):
        """Per-call HIP events (last_timing) on / off; off: nothing but the kernels goes on the stream (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_set_timing(self._ctx, 1 if on else 0))

    def set_msm_lanes(self, lanes=2):
        """Two MSMs in flight on an asynchronous context (alternating internal streams and workspaces); their outputs are
        ordered by synchronize() only (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_set_msm_lanes(self._ctx, int(lanes)))

    def last_timing(self, name="total"):
        ms = ctypes.c_double(0)
        rc = self._lib.ecgpu_last_timing(self._ctx, name.encode(), ctypes.byref(ms))
        return ms.value if rc == OK else None

    def valu_probe(self, which=0):
        v = ctypes.c_double(0)
        self._chk(self._lib.ecgpu_valu_probe(self._ctx, which, ctypes.byref(v)))
        return v.value

    # ---- host-buffer operations (numpy uint8 / bytes in, numpy out) ----
    def mul_by_generator(self, curve, scalars, out=None, inf=None, constant_time=False):
        """out / inf: optional preallocated uint8 arrays (n*2L, n), e.g. from host_alloc.
        constant_time: the uniform-schedule entry point (ecgpu_batch_mul_base_ct) for secret scalars."""
        L = _field_bytes(curve)
        s = _host(scalars)
        n = s.size // L
        out = np.zeros(n * 2 * L, np.uint8) if out is None else out
        inf = np.zeros(n, np.uint8) if inf is None else inf
        assert out.dtype == np.uint8 and out.size >= n * 2 * L and inf.size >= n and out.flags.c_contiguous
        fn = self._lib.ecgpu_batch_mul_base_ct if constant_time else self._lib.ecgpu_batch_mul_base
        self._chk(fn(self._ctx, curve, _hp(s), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def mul_by_generator_compressed(self, curve, scalars, out_x=None, out_tag=None):
        """k_i * G in SEC1-compressed form: (x uint8[n*L], tag uint8[n] = 2 / 3, 0 for the identity)."""
        L = _field_bytes(curve)
        s = _host(scalars)
        n = s.size // L
        out_x = np.zeros(n * L, np.uint8) if out_x is None else out_x
        out_tag = np.zeros(n, np.uint8) if out_tag is None else out_tag
        for o, need in ((out_x, n * L), (out_tag, n)):
            if o.dtype != np.uint8 or not o.flags.c_contiguous or o.size < need:
                raise EcgpuError(ERR_ARG, "output buffer must be contiguous uint8 of at least %d bytes" % need)
        self._chk(self._lib.ecgpu_batch_mul_base_compressed(self._ctx, curve, _hp(s), ctypes.c_size_t(n), _hp(out_x), _hp(out_tag)))
        return out_x, out_tag

    def mul_by_generator_compressed_dev(self, curve, d_scalars, n, d_out_x, d_out_tag):
        self._chk(self._lib.ecgpu_batch_mul_base_compressed_dev(self._ctx, curve, _dp(d_scalars), ctypes.c_size_t(n), _dp(d_out_x),
                                                                _dp(d_out_tag)))

    def mul(self, curve, scalars, points_xy, points_inf=None, constant_time=False):
        """k_i * P_i.  constant_time: the uniform-schedule entry point (ecgpu_batch_mul_ct) for secret scalars."""
        L = _field_bytes(curve)
        s, p, pi = _host(scalars), _host(points_xy), _host(points_inf)
        n = s.size // L
        _need("scalars", s, n * L); _need("points_xy", p, n * 2 * L); _need("points_inf", pi, n)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        fn = self._lib.ecgpu_batch_mul_ct if constant_time else self._lib.ecgpu_batch_mul
        self._chk(fn(self._ctx, curve, _hp(s), _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def lincomb(self, curve, scalars, points_xy, points_inf=None):
        L = _field_bytes(curve)
        s, p, pi = _host(scalars), _host(points_xy), _host(points_inf)
        n = s.size // L
        _need("scalars", s, n * L); _need("points_xy", p, n * 2 * L); _need("points_inf", pi, n)
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_msm(self._ctx, curve, _hp(s), _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, int(inf[0])

    def lincomb_ct(self, curve, scalars, points_xy, points_inf=None):
        """sum_i k_i P_i by the uniform-schedule entry point (ecgpu_lincomb_ct: `LinearCombination::lincomb` in its constant-time
        form) — for secret scalars; `lincomb` (the bucket method) is the one for public data."""
        L = _field_bytes(curve)
        s, p, pi = _host(scalars), _host(points_xy), _host(points_inf)
        n = s.size // L
        _need("scalars", s, n * L); _need("points_xy", p, n * 2 * L); _need("points_inf", pi, n)
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_lincomb_ct(self._ctx, curve, _hp(s), _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, int(inf[0])

    def lincomb_compressed(self, curve, scalars, points_x, points_tag):
        """sum_i k_i P_i with SEC1-compressed points: points_x n*L bytes, points_tag n bytes (0x02 / 0x03, 0x00 = identity)."""
        L = _field_bytes(curve)
        s, x, t = _host(scalars), _host(points_x), _host(points_tag)
        n = s.size // L
        _need("scalars", s, n * L); _need("points_x", x, n * L); _need("points_tag", t, n)
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_msm_compressed(self._ctx, curve, _hp(s), _hp(x), _hp(t), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, int(inf[0])

    def mul_compressed(self, curve, scalars, points_x, points_tag):
        """k_i * P_i with SEC1-compressed points (see lincomb_compressed); returns (xy uint8[n*2L], inf uint8[n])."""
        L = _field_bytes(curve)
        s, x, t = _host(scalars), _host(points_x), _host(points_tag)
        n = s.size // L
        _need("scalars", s, n * L); _need("points_x", x, n * L); _need("points_tag", t, n)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_batch_mul_compressed(self._ctx, curve, _hp(s), _hp(x), _hp(t), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def wipe(self):
        """ecgpu_wipe: zero every staging / scratch buffer of the context on the device."""
        self._chk(self._lib.ecgpu_wipe(self._ctx))

    def mul_by_generator_and_mul_add(self, curve, a_scalars, b_scalars, points_xy, points_inf=None):
        L = _field_bytes(curve)
        a, b, p, pi = _host(a_scalars), _host(b_scalars), _host(points_xy), _host(points_inf)
        n = a.size // L
        _need("a_scalars", a, n * L); _need("b_scalars", b, n * L); _need("points_xy", p, n * 2 * L); _need("points_inf", pi, n)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_batch_mul_base_and_mul_add(self._ctx, curve, _hp(a), _hp(b), _hp(p), _hp(pi),
                                                             ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def ecdsa_verify(self, curve, z, r, s, q_xy, reject_high_s=False):
        """Batch ECDSA verification: z, r, s are n*L big-endian bytes each, q_xy n*2L; returns uint8[n] (1 = valid)."""
        L = _field_bytes(curve)
        zz, rr, ss, qq = _host(z), _host(r), _host(s), _host(q_xy)
        n = zz.size // L
        _need("z", zz, n * L); _need("r", rr, n * L); _need("s", ss, n * L); _need("q_xy", qq, n * 2 * L)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_ecdsa_verify_batch(self._ctx, curve, _hp(zz), _hp(rr), _hp(ss), _hp(qq), ctypes.c_size_t(n),
                                                     int(bool(reject_high_s)), _hp(ok)))
        return ok

    def ecdsa_recover(self, curve, z, r, s, recid, reject_high_s=False):
        """Batch ECDSA public-key recovery: z, r, s n*L big-endian bytes each, recid n recovery id bytes (bit 0: y(R) odd,
        bit 1: x(R) = r + n); returns (keys uint8[n*2L] — zero records where recovery fails —, ok uint8[n])."""
        L = _field_bytes(curve)
        zz, rr, ss, ii = _host(z), _host(r), _host(s), _host(recid)
        n = ii.size
        _need("z", zz, n * L); _need("r", rr, n * L); _need("s", ss, n * L)
        out = np.zeros(n * 2 * L, np.uint8)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_ecdsa_recover_batch(self._ctx, curve, _hp(zz), _hp(rr), _hp(ss), _hp(ii), ctypes.c_size_t(n),
                                                      int(bool(reject_high_s)), _hp(out), _hp(ok)))
        return out, ok

    def schnorr_verify(self, e, r, s, p_xy):
        """Batch BIP340 verification (k256): e = challenge hash as 32 bytes, (r, s) signature halves, p_xy lifted key."""
        ee, rr, ss, pp = _host(e), _host(r), _host(s), _host(p_xy)
        n = ee.size // 32
        _need("e", ee, n * 32); _need("r", rr, n * 32); _need("s", ss, n * 32); _need("p_xy", pp, n * 64)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_schnorr_verify_batch(self._ctx, _hp(ee), _hp(rr), _hp(ss), _hp(pp), ctypes.c_size_t(n), _hp(ok)))
        return ok

    def sm2dsa_verify(self, e, r, s, q_xy):
        """Batch SM2DSA verification on the prehash (sm2): e = SM3(ZA || M) as 32 bytes, (r, s), the public keys."""
        ee, rr, ss, qq = _host(e), _host(r), _host(s), _host(q_xy)
        n = ee.size // 32
        _need("e", ee, n * 32); _need("r", rr, n * 32); _need("s", ss, n * 32); _need("q_xy", qq, n * 64)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_sm2dsa_verify_batch(self._ctx, _hp(ee), _hp(rr), _hp(ss), _hp(qq), ctypes.c_size_t(n), _hp(ok)))
        return ok

    def ecdsa_verify_msg(self, curve, q_xy, msgs, msg_len, sigs, reject_high_s=False):
        """Batch ECDSA verification of messages: keys n*2L, messages n*msg_len, signatures n*2L (r || s); the curve's digest
        (SHA-256 / 384 / 224 / 512) and bits2field run on the device."""
        L = _field_bytes(curve)
        qq, sg = _host(q_xy), _host(sigs)
        mm = _host(msgs) if msg_len else None
        n = qq.size // (2 * L)
        _need("q_xy", qq, n * 2 * L); _need("sigs", sg, n * 2 * L); _need("msgs", mm, n * msg_len)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_ecdsa_verify_msg_batch(self._ctx, curve, _hp(qq), _hp(mm), ctypes.c_size_t(msg_len), _hp(sg),
                                                         ctypes.c_size_t(n), int(bool(reject_high_s)), _hp(ok)))
        return ok

    def sm2dsa_verify_msg(self, distid, q_xy, msgs, msg_len, sigs):
        """Batch SM2DSA verification of messages (sm2): one distinguishing identifier, keys n*64, messages n*msg_len, signatures
        n*64 (r || s); Z and e = SM3(Z || M) are computed on the device."""
        dd, qq, sg = _host(distid) if len(distid) else None, _host(q_xy), _host(sigs)
        mm = _host(msgs) if msg_len else None
        n = qq.size // 64
        _need("q_xy", qq, n * 64); _need("sigs", sg, n * 64); _need("msgs", mm, n * msg_len)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_sm2dsa_verify_msg_batch(self._ctx, _hp(dd), ctypes.c_size_t(len(distid)), _hp(qq), _hp(mm),
                                                          ctypes.c_size_t(msg_len), _hp(sg), ctypes.c_size_t(n), _hp(ok)))
        return ok

    def bign_verify(self, h, sigs, q_xy):
        """Batch bign verification on the prehash (bign256): h = belt-hash(message) as 32 bytes, signatures n*48 (S0 || S1), keys
        n*64; all little-endian."""
        hh, sg, qq = _host(h), _host(sigs), _host(q_xy)
        n = hh.size // 32
        _need("h", hh, n * 32); _need("sigs", sg, n * 48); _need("q_xy", qq, n * 64)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_bign_verify_batch(self._ctx, _hp(hh), _hp(sg), _hp(qq), ctypes.c_size_t(n), _hp(ok)))
        return ok

    def bign_verify_msg(self, q_xy, msgs, msg_len, sigs):
        """Batch bign verification of messages (bign256): keys n*64, messages n*msg_len, signatures n*48; belt-hash runs on the
        device."""
        qq, sg = _host(q_xy), _host(sigs)
        mm = _host(msgs) if msg_len else None
        n = qq.size // 64
        _need("q_xy", qq, n * 64); _need("sigs", sg, n * 48); _need("msgs", mm, n * msg_len)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_bign_verify_msg_batch(self._ctx, _hp(qq), _hp(mm), ctypes.c_size_t(msg_len), _hp(sg),
                                                        ctypes.c_size_t(n), _hp(ok)))
        return ok

    def ecdh(self, curve, scalars, points_xy, constant_time=False):
        """x-coordinates of k_i * P_i (ECDH shared secrets): returns (x uint8[n*L], ok uint8[n]).
        constant_time: the uniform-schedule entry point (ecgpu_batch_ecdh_ct) for secret scalars."""
        L = _field_bytes(curve)
        k, p = _host(scalars), _host(points_xy)
        n = k.size // L
        _need("scalars", k, n * L); _need("points_xy", p, n * 2 * L)
        out = np.zeros(n * L, np.uint8)
        ok = np.zeros(n, np.uint8)
        fn = self._lib.ecgpu_batch_ecdh_ct if constant_time else self._lib.ecgpu_batch_ecdh
        self._chk(fn(self._ctx, curve, _hp(k), _hp(p), ctypes.c_size_t(n), _hp(out), _hp(ok)))
        return out, ok

    def ecdh_dev(self, curve, d_scalars, d_points_xy, n, d_out_x, d_ok, constant_time=False):
        fn = self._lib.ecgpu_batch_ecdh_ct_dev if constant_time else self._lib.ecgpu_batch_ecdh_dev
        self._chk(fn(self._ctx, curve, _dp(d_scalars), _dp(d_points_xy), ctypes.c_size_t(n), _dp(d_out_x), _dp(d_ok)))

    def schnorr_verify_raw(self, pk_x, msgs, msg_len, sigs):
        """BIP340 verification from wire bytes: x-only keys (n*32), messages (n*msg_len), signatures (n*64)."""
        pk, mm, sg = _host(pk_x), _host(msgs), _host(sigs)
        n = pk.size // 32
        _need("pk_x", pk, n * 32); _need("sigs", sg, n * 64)
        if msg_len:
            _need("msgs", mm, n * msg_len)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_schnorr_verify_raw_batch(self._ctx, _hp(pk), _hp(mm) if msg_len else None,
                                                           ctypes.c_size_t(msg_len), _hp(sg), ctypes.c_size_t(n), _hp(ok)))
        return ok

    def schnorr_verify_raw_dev(self, d_pk_x, d_msgs, msg_len, d_sigs, n, d_ok):
        self._chk(self._lib.ecgpu_schnorr_verify_raw_batch_dev(self._ctx, _dp(d_pk_x), _dp(d_msgs) if msg_len else None,
                                                               ctypes.c_size_t(msg_len), _dp(d_sigs), ctypes.c_size_t(n), _dp(d_ok)))

    def decompress(self, curve, xs, y_is_odd):
        """DecompressPoint::decompress for a batch: returns (xy uint8[n*2L], ok uint8[n])."""
        L = _field_bytes(curve)
        x, odd = _host(xs), _host(y_is_odd)
        n = x.size // L
        _need("xs", x, n * L); _need("y_is_odd", odd, n)
        out = np.zeros(n * 2 * L, np.uint8)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_batch_decompress(self._ctx, curve, _hp(x), _hp(odd), ctypes.c_size_t(n), _hp(out), _hp(ok)))
        return out, ok

    def batch_normalize(self, curve, points_xyz):
        L = _field_bytes(curve)
        x = _host(points_xyz)
        n = x.size // (3 * L)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_batch_normalize(self._ctx, curve, _hp(x), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def point_sum(self, curve, points_xy, points_inf=None):
        L = _field_bytes(curve)
        p, pi = _host(points_xy), _host(points_inf)
        n = p.size // (2 * L)
        _need("points_xy", p, n * 2 * L); _need("points_inf", pi, n)
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_point_sum(self._ctx, curve, _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, int(inf[0])

    def k256_glv_decompose(self, scalars):
        s = _host(scalars)
        n = s.size // 32
        r1 = np.zeros(n * 32, np.uint8)
        r2 = np.zeros(n * 32, np.uint8)
        self._chk(self._lib.ecgpu_k256_glv_decompose(self._ctx, _hp(s), ctypes.c_size_t(n), _hp(r1), _hp(r2)))
        return r1, r2

    # ---- device-side known-answer tests of the field / group arithmetic (ecgpu_selftest_*) ----
    def selftest_field(self, curve, op, a, b=None):
        L = _field_bytes(curve)
        A, B = _host(a), _host(b)
        n = A.size // L
        _need("a", A, n * L); _need("b", B, n * L)
        out = np.zeros(n * L, np.uint8)
        self._chk(self._lib.ecgpu_selftest_field(self._ctx, curve, int(op), _hp(A), _hp(B), ctypes.c_size_t(n), _hp(out)))
        return out

    def selftest_point(self, curve, op, p_xy, p_inf=None, q_xy=None, q_inf=None):
        L = _field_bytes(curve)
        P, PI, Q, QI = _host(p_xy), _host(p_inf), _host(q_xy), _host(q_inf)
        n = P.size // (2 * L)
        _need("p_xy", P, n * 2 * L); _need("p_inf", PI, n); _need("q_xy", Q, n * 2 * L); _need("q_inf", QI, n)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_selftest_point(self._ctx, curve, int(op), _hp(P), _hp(PI), _hp(Q), _hp(QI), ctypes.c_size_t(n),
                                                 _hp(out), _hp(inf)))
        return out, inf

    # ---- device-resident operations (torch uint8 CUDA tensors or raw device pointers) ----
    def mul_by_generator_dev(self, curve, d_scalars, n, d_out_xy, d_out_inf=None, constant_time=False):
        fn = self._lib.ecgpu_batch_mul_base_ct_dev if constant_time else self._lib.ecgpu_batch_mul_base_dev
        self._chk(fn(self._ctx, curve, _dp(d_scalars), ctypes.c_size_t(n), _dp(d_out_xy), _dp(d_out_inf)))

    def mul_dev(self, curve, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf=None, constant_time=False):
        fn = self._lib.ecgpu_batch_mul_ct_dev if constant_time else self._lib.ecgpu_batch_mul_dev
        self._chk(fn(self._ctx, curve, _dp(d_scalars), _dp(d_points_xy), _dp(d_points_inf), ctypes.c_size_t(n), _dp(d_out_xy),
                     _dp(d_out_inf)))

    def lincomb_dev(self, curve, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf):
        self._chk(self._lib.ecgpu_msm_dev(self._ctx, curve, _dp(d_scalars), _dp(d_points_xy), _dp(d_points_inf),
                                          ctypes.c_size_t(n), _dp(d_out_xy), _dp(d_out_inf)))

    def lincomb_ct_dev(self, curve, d_scalars, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf):
        self._chk(self._lib.ecgpu_lincomb_ct_dev(self._ctx, curve, _dp(d_scalars), _dp(d_points_xy), _dp(d_points_inf),
                                                 ctypes.c_size_t(n), _dp(d_out_xy), _dp(d_out_inf)))

    def lincomb_compressed_dev(self, curve, d_scalars, d_points_x, d_points_tag, n, d_out_xy, d_out_inf):
        self._chk(self._lib.ecgpu_msm_compressed_dev(self._ctx, curve, _dp(d_scalars), _dp(d_points_x), _dp(d_points_tag),
                                                     ctypes.c_size_t(n), _dp(d_out_xy), _dp(d_out_inf)))

    def mul_compressed_dev(self, curve, d_scalars, d_points_x, d_points_tag, n, d_out_xy, d_out_inf=None):
        self._chk(self._lib.ecgpu_batch_mul_compressed_dev(self._ctx, curve, _dp(d_scalars), _dp(d_points_x), _dp(d_points_tag),
                                                           ctypes.c_size_t(n), _dp(d_out_xy), _dp(d_out_inf)))

    # an MSM spread over several GPUs: local half -> all-gather of the parts -> combining half (include/ecgpu.h)
    def msm_parts_bytes(self, curve, plan_terms):
        return int(self._lib.ecgpu_msm_parts_bytes(self._ctx, curve, ctypes.c_size_t(plan_terms)))

    def msm_parts_dev(self, curve, d_scalars, d_points_xy, d_points_inf, n, plan_terms, d_parts):
        self._chk(self._lib.ecgpu_msm_parts_dev(self._ctx, curve, _dp(d_scalars), _dp(d_points_xy), _dp(d_points_inf),
                                                ctypes.c_size_t(n), ctypes.c_size_t(plan_terms), _dp(d_parts)))

    def msm_parts_join_dev(self, d_parts):
        """The context's stream waits (on the device) for the local half that wrote `d_parts` on a lane (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_msm_parts_join_dev(self._ctx, _dp(d_parts)))

    def msm_finish_dev(self, curve, d_parts_all, nranks, plan_terms, d_out_xy, d_out_inf):
        self._chk(self._lib.ecgpu_msm_finish_dev(self._ctx, curve, _dp(d_parts_all), int(nranks), ctypes.c_size_t(plan_terms),
                                                 _dp(d_out_xy), _dp(d_out_inf)))

    def ecdsa_verify_dev(self, curve, d_z, d_r, d_s, d_q_xy, n, reject_high_s, d_ok):
        self._chk(self._lib.ecgpu_ecdsa_verify_batch_dev(self._ctx, curve, _dp(d_z), _dp(d_r), _dp(d_s), _dp(d_q_xy),
                                                         ctypes.c_size_t(n), int(bool(reject_high_s)), _dp(d_ok)))

    def ecdsa_verify_msg_dev(self, curve, d_q_xy, d_msgs, msg_len, d_sigs, n, reject_high_s, d_ok):
        self._chk(self._lib.ecgpu_ecdsa_verify_msg_batch_dev(self._ctx, curve, _dp(d_q_xy), _dp(d_msgs), ctypes.c_size_t(msg_len),
                                                             _dp(d_sigs), ctypes.c_size_t(n), int(bool(reject_high_s)), _dp(d_ok)))

    def sm2dsa_verify_msg_dev(self, d_distid, distid_len, d_q_xy, d_msgs, msg_len, d_sigs, n, d_ok):
        self._chk(self._lib.ecgpu_sm2dsa_verify_msg_batch_dev(self._ctx, _dp(d_distid), ctypes.c_size_t(distid_len), _dp(d_q_xy),
                                                              _dp(d_msgs), ctypes.c_size_t(msg_len), _dp(d_sigs), ctypes.c_size_t(n),
                                                              _dp(d_ok)))

    def bign_verify_dev(self, d_h, d_sigs, d_q_xy, n, d_ok):
        self._chk(self._lib.ecgpu_bign_verify_batch_dev(self._ctx, _dp(d_h), _dp(d_sigs), _dp(d_q_xy), ctypes.c_size_t(n), _dp(d_ok)))

    def bign_verify_msg_dev(self, d_q_xy, d_msgs, msg_len, d_sigs, n, d_ok):
        self._chk(self._lib.ecgpu_bign_verify_msg_batch_dev(self._ctx, _dp(d_q_xy), _dp(d_msgs), ctypes.c_size_t(msg_len), _dp(d_sigs),
                                                            ctypes.c_size_t(n), _dp(d_ok)))

    def ecdsa_recover_dev(self, curve, d_z, d_r, d_s, d_recid, n, reject_high_s, d_out_xy, d_ok):
        self._chk(self._lib.ecgpu_ecdsa_recover_batch_dev(self._ctx, curve, _dp(d_z), _dp(d_r), _dp(d_s), _dp(d_recid),
                                                          ctypes.c_size_t(n), int(bool(reject_high_s)), _dp(d_out_xy), _dp(d_ok)))

    def point_sum_dev(self, curve, d_points_xy, d_points_inf, n, d_out_xy, d_out_inf):
        self._chk(self._lib.ecgpu_point_sum_dev(self._ctx, curve, _dp(d_points_xy), _dp(d_points_inf), ctypes.c_size_t(n),
                                                _dp(d_out_xy), _dp(d_out_inf)))


class Group:
    """All GPUs of a node from one process (ecgpu_group_*): batch calls slice the index range, lincomb shards its terms
    and exchanges per-window partial sums once."""

    def __init__(self, devices, variant=None, exchange=None):
        """exchange: None (RCCL when it can be had, else peer copies), "peer", or "rccl" (EcgpuError instead of the fallback)."""
        self._lib = load_library(variant)
        self._g = ctypes.c_void_p()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.ecgpu_group_init(ctypes.byref(self._g), devs, len(devices))
        if rc != OK:
            self._g = None
            raise EcgpuError(rc, "ecgpu_group_init failed")
        self.size = int(self._lib.ecgpu_group_size(self._g))
        if exchange is not None:
            try:
                self.set_exchange(exchange)
            except EcgpuError:
                self.close()
                raise

    def set_exchange(self, mode):
        """ecgpu_group_set_exchange: "peer" = peer copies from now on; "rccl" = raise unless the group exchanges over RCCL."""
        self._chk(self._lib.ecgpu_group_set_exchange(self._g, {"peer": EXCHANGE_PEER, "rccl": EXCHANGE_RCCL}[mode]))

    @property
    def exchange(self):
        """"rccl" or "peer" — as of now: a collective that failed or missed its deadline moves the group to peer copies."""
        return self._lib.ecgpu_group_exchange(self._g).decode()

    @property
    def exchange_reason(self):
        return self._lib.ecgpu_group_exchange_reason(self._g).decode()

    def close(self):
        if getattr(self, "_g", None):
            self._lib.ecgpu_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise EcgpuError(rc, (self._lib.ecgpu_group_last_error(self._g) or b"").decode())

    def set_exchange_timeout(self, seconds):
        """The MSM's exchange step gives a collective up after this long and completes over peer copies (include/ecgpu.h)."""
        self._chk(self._lib.ecgpu_group_set_exchange_timeout(self._g, ctypes.c_double(float(seconds))))

    def set_msm_window(self, bits):
        self._chk(self._lib.ecgpu_group_set_msm_window(self._g, int(bits)))

    def lincomb(self, curve, scalars, points_xy, points_inf=None):
        L = _field_bytes(curve)
        s, p, pi = _host(scalars), _host(points_xy), _host(points_inf)
        n = s.size // L
        if p.size != n * 2 * L or (pi is not None and pi.size != n):
            raise EcgpuError(ERR_ARG, "lincomb: buffer sizes do not match %d terms" % n)
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_group_msm(self._g, curve, _hp(s), _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, int(inf[0])

    def lincomb_dev(self, curve, d_scalars, d_points_xy, n_per_device, d_points_inf=None):
        """ecgpu_group_msm_dev: shard i (n_per_device[i] terms) already resident on member i's device — lists of torch
        tensors / DeviceBuffers / None, one per member.  Returns (xy uint8[2L], inf) on the host."""
        L = _field_bytes(curve)
        m = self.size
        if not (len(d_scalars) == len(d_points_xy) == len(n_per_device) == m) or (d_points_inf is not None and len(d_points_inf) != m):
            raise EcgpuError(ERR_ARG, "lincomb_dev: one entry per group member expected")
        arr = lambda xs: (ctypes.c_void_p * m)(*[(_dp(x).value if x is not None else None) for x in xs])
        ds, dp = arr(d_scalars), arr(d_points_xy)
        di = arr(d_points_inf) if d_points_inf is not None else None
        cnt = (ctypes.c_size_t * m)(*[int(v) for v in n_per_device])
        out = np.zeros(2 * L, np.uint8)
        inf = np.zeros(1, np.uint8)
        self._chk(self._lib.ecgpu_group_msm_dev(self._g, curve, ds, dp, di, cnt, _hp(out), _hp(inf)))
        return out, int(inf[0])

    def mul_by_generator(self, curve, scalars):
        L = _field_bytes(curve)
        s = _host(scalars)
        n = s.size // L
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_group_batch_mul_base(self._g, curve, _hp(s), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def mul(self, curve, scalars, points_xy, points_inf=None):
        L = _field_bytes(curve)
        s, p, pi = _host(scalars), _host(points_xy), _host(points_inf)
        n = s.size // L
        if p.size != n * 2 * L or (pi is not None and pi.size != n):
            raise EcgpuError(ERR_ARG, "mul: buffer sizes do not match %d units" % n)
        out = np.zeros(n * 2 * L, np.uint8)
        inf = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_group_batch_mul(self._g, curve, _hp(s), _hp(p), _hp(pi), ctypes.c_size_t(n), _hp(out), _hp(inf)))
        return out, inf

    def ecdsa_verify(self, curve, z, r, s, q_xy, reject_high_s=False):
        L = _field_bytes(curve)
        zz, rr, ss, qq = _host(z), _host(r), _host(s), _host(q_xy)
        n = zz.size // L
        _need("r", rr, n * L); _need("s", ss, n * L); _need("q_xy", qq, n * 2 * L)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_group_ecdsa_verify_batch(self._g, curve, _hp(zz), _hp(rr), _hp(ss), _hp(qq), ctypes.c_size_t(n),
                                                           int(bool(reject_high_s)), _hp(ok)))
        return ok

    def ecdsa_verify_msg(self, curve, q_xy, msgs, msg_len, sigs, reject_high_s=False):
        L = _field_bytes(curve)
        qq, sg = _host(q_xy), _host(sigs)
        mm = _host(msgs) if msg_len else None
        n = qq.size // (2 * L)
        _need("sigs", sg, n * 2 * L); _need("msgs", mm, n * msg_len)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_group_ecdsa_verify_msg_batch(self._g, curve, _hp(qq), _hp(mm), ctypes.c_size_t(msg_len), _hp(sg),
                                                               ctypes.c_size_t(n), int(bool(reject_high_s)), _hp(ok)))
        return ok

    def ecdsa_recover(self, curve, z, r, s, recid, reject_high_s=False):
        L = _field_bytes(curve)
        zz, rr, ss, ii = _host(z), _host(r), _host(s), _host(recid)
        n = ii.size
        _need("z", zz, n * L); _need("r", rr, n * L); _need("s", ss, n * L)
        out = np.zeros(n * 2 * L, np.uint8)
        ok = np.zeros(n, np.uint8)
        self._chk(self._lib.ecgpu_group_ecdsa_recover_batch(self._g, curve, _hp(zz), _hp(rr), _hp(ss), _hp(ii), ctypes.c_size_t(n),
                                                            int(bool(reject_high_s)), _hp(out), _hp(ok)))
        return out, ok



def version():
    return load_library().ecgpu_version().decode()


from .sharded import Exchange, LocalRecord, RecordExchange, TensorExchange, init_exchange, lincomb_sharded, run_nccl_probe, shard_range  # noqa: E402,F401
