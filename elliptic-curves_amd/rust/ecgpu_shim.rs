//! ecgpu_shim.rs — the reference-side binding a maintainer would add (NOT built in this repo: the
//! image has no rustc/cargo; kept next to include/ecgpu.h and checked against it by hand).
//!
//! It exposes libecgpu.so behind the reference's own plug-in points:
//!   * `primeorder::MulBackend<C>`            (primeorder/src/mul_backend.rs:11-40)
//!   * `elliptic_curve::ops::LinearCombination` (primeorder/src/projective.rs:480-511,
//!                                               k256/src/arithmetic/mul.rs:84-109)
//!   * new batch entry points `batch_mul_by_generator`, `batch_mul`
//!
//! Build: add `links = "ecgpu"` + a build.rs emitting `cargo:rustc-link-lib=dylib=ecgpu` and
//! `cargo:rustc-link-search=<repo>/elliptic-curves_amd/lib`.

use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct EcgpuCtx {
    _private: [u8; 0],
}

pub const ECGPU_K256: c_int = 0;
pub const ECGPU_P256: c_int = 1;
pub const ECGPU_P384: c_int = 2;
pub const ECGPU_SM2: c_int = 3;
pub const ECGPU_P224: c_int = 4;
pub const ECGPU_P192: c_int = 5;
pub const ECGPU_P521: c_int = 6;
pub const ECGPU_BP256: c_int = 7;
pub const ECGPU_BP384: c_int = 8;
pub const ECGPU_BP256T1: c_int = 9;
pub const ECGPU_BP384T1: c_int = 10;

pub const ECGPU_OK: c_int = 0;
pub const ECGPU_ERR_SCALAR_RANGE: c_int = -2;
pub const ECGPU_ERR_POINT: c_int = -3;
pub const ECGPU_ERR_NO_DEVICE: c_int = -4;

#[link(name = "ecgpu")]
unsafe extern "C" {
    pub fn ecgpu_init(ctx: *mut *mut EcgpuCtx, device: c_int) -> c_int;
    pub fn ecgpu_destroy(ctx: *mut EcgpuCtx);
    pub fn ecgpu_last_error(ctx: *const EcgpuCtx) -> *const c_char;
    pub fn ecgpu_field_bytes(curve: c_int) -> usize;
    pub fn ecgpu_set_stream(ctx: *mut EcgpuCtx, stream: *mut c_void) -> c_int;
    pub fn ecgpu_host_alloc(ctx: *mut EcgpuCtx, bytes: usize) -> *mut c_void;
    pub fn ecgpu_host_free(ctx: *mut EcgpuCtx, p: *mut c_void);
    pub fn ecgpu_set_base_window(ctx: *mut EcgpuCtx, curve: c_int, window_bits: c_int) -> c_int;
    pub fn ecgpu_set_msm_window(ctx: *mut EcgpuCtx, window_bits: c_int) -> c_int;
    pub fn ecgpu_batch_mul_base(ctx: *mut EcgpuCtx, curve: c_int, scalars: *const u8, n: usize,
                                out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn ecgpu_batch_mul_base_compressed(ctx: *mut EcgpuCtx, curve: c_int, scalars: *const u8, n: usize,
                                           out_x: *mut u8, out_tag: *mut u8) -> c_int;
    pub fn ecgpu_batch_mul_base_compressed_dev(ctx: *mut EcgpuCtx, curve: c_int, d_scalars: *const c_void, n: usize,
                                               d_out_x: *mut c_void, d_out_tag: *mut c_void) -> c_int;
    pub fn ecgpu_batch_mul(ctx: *mut EcgpuCtx, curve: c_int, scalars: *const u8, points_xy: *const u8,
                           points_inf: *const u8, n: usize, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn ecgpu_msm(ctx: *mut EcgpuCtx, curve: c_int, scalars: *const u8, points_xy: *const u8,
                     points_inf: *const u8, n: usize, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn ecgpu_batch_mul_base_and_mul_add(ctx: *mut EcgpuCtx, curve: c_int, a: *const u8, b: *const u8,
                                            points_xy: *const u8, points_inf: *const u8, n: usize,
                                            out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn ecgpu_batch_normalize(ctx: *mut EcgpuCtx, curve: c_int, points_xyz: *const u8, n: usize,
                                 out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    pub fn ecgpu_point_sum(ctx: *mut EcgpuCtx, curve: c_int, points_xy: *const u8, points_inf: *const u8,
                           n: usize, out_xy: *mut u8, out_inf: *mut u8) -> c_int;
    /// Batch form of `ecdsa::hazmat::verify_prehashed`; `reject_high_s` = `C::NORMALIZE_S`.
    pub fn ecgpu_ecdsa_verify_batch(ctx: *mut EcgpuCtx, curve: c_int, z: *const u8, r: *const u8, s: *const u8,
                                    q_xy: *const u8, n: usize, reject_high_s: c_int, ok: *mut u8) -> c_int;
    /// Batch form of `schnorr::VerifyingKey::verify_raw` (k256); `e` is the BIP340 challenge hash.
    pub fn ecgpu_schnorr_verify_batch(ctx: *mut EcgpuCtx, e: *const u8, r: *const u8, s: *const u8, p_xy: *const u8,
                                      n: usize, ok: *mut u8) -> c_int;
    /// `VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)` for a batch of equally long messages.
    pub fn ecgpu_schnorr_verify_raw_batch(ctx: *mut EcgpuCtx, pk_x: *const u8, msgs: *const u8, msg_len: usize,
                                          sigs: *const u8, n: usize, ok: *mut u8) -> c_int;
    /// Batch form of `elliptic_curve::ecdh::diffie_hellman`: x-coordinates of k_i * P_i.
    pub fn ecgpu_batch_ecdh(ctx: *mut EcgpuCtx, curve: c_int, scalars: *const u8, points_xy: *const u8, n: usize,
                            out_x: *mut u8, ok: *mut u8) -> c_int;
    /// Batch form of `DecompressPoint::decompress(x_bytes, y_is_odd)`.
    pub fn ecgpu_batch_decompress(ctx: *mut EcgpuCtx, curve: c_int, xs: *const u8, y_is_odd: *const u8, n: usize,
                                  out_xy: *mut u8, ok: *mut u8) -> c_int;
}

/// Process-wide context: the analogue of `static BASEPOINT_TABLE: LazyLock<..>`
/// (k256/src/arithmetic/tables.rs:18).
pub struct Engine(*mut EcgpuCtx);
unsafe impl Send for Engine {}
unsafe impl Sync for Engine {}

pub static ENGINE: std::sync::LazyLock<std::sync::Mutex<Engine>> = std::sync::LazyLock::new(|| {
    let mut ctx = core::ptr::null_mut();
    let rc = unsafe { ecgpu_init(&mut ctx, 0) };
    assert_eq!(rc, ECGPU_OK, "no gfx950 device: the GPU backend has no CPU fallback");
    std::sync::Mutex::new(Engine(ctx))
});

// ---- p256: a `MulBackend` that a curve crate selects via `PrimeCurveParams::Backend` ----------------
// (primeorder/src/lib.rs:62; compare p256/src/arithmetic/tables.rs:24-44)
pub mod p256_backend {
    use super::*;
    use elliptic_curve::{
        ops::LinearCombination,
        point::AffineCoordinates,
        sec1::{FromSec1Point, ToSec1Point},
        PrimeField,
    };
    use p256::{AffinePoint, NistP256, ProjectivePoint, Scalar};
    use primeorder::MulBackend;

    fn to_wire(p: &ProjectivePoint) -> ([u8; 64], u8) {
        let a = p.to_affine();
        let mut xy = [0u8; 64];
        if bool::from(a.is_identity()) {
            return (xy, 1);
        }
        xy[..32].copy_from_slice(&a.x());
        xy[32..].copy_from_slice(&a.y());
        (xy, 0)
    }

    fn from_wire(xy: &[u8], inf: u8) -> ProjectivePoint {
        if inf != 0 {
            return ProjectivePoint::IDENTITY;
        }
        let x = p256::FieldBytes::try_from(&xy[..32]).unwrap();
        let y = p256::FieldBytes::try_from(&xy[32..64]).unwrap();
        ProjectivePoint::from(AffinePoint::from_coordinates(&x, &y).unwrap())
    }

    /// New API: `k[i] * G` for a whole slice on the GPU.
    pub fn batch_mul_by_generator(ks: &[Scalar]) -> Vec<ProjectivePoint> {
        let scalars: Vec<u8> = ks.iter().flat_map(|k| k.to_repr()).collect();
        let mut xy = vec![0u8; ks.len() * 64];
        let mut inf = vec![0u8; ks.len()];
        let eng = ENGINE.lock().unwrap();
        let rc = unsafe {
            ecgpu_batch_mul_base(eng.0, ECGPU_P256, scalars.as_ptr(), ks.len(), xy.as_mut_ptr(), inf.as_mut_ptr())
        };
        assert_eq!(rc, ECGPU_OK);
        xy.chunks(64).zip(inf).map(|(c, f)| from_wire(c, f)).collect()
    }

    /// New API: `k[i] * P[i]`.
    pub fn batch_mul(terms: &[(ProjectivePoint, Scalar)]) -> Vec<ProjectivePoint> {
        let n = terms.len();
        let scalars: Vec<u8> = terms.iter().flat_map(|(_, k)| k.to_repr()).collect();
        let (mut pts, mut pinf) = (Vec::with_capacity(n * 64), Vec::with_capacity(n));
        for (p, _) in terms {
            let (xy, f) = to_wire(p);
            pts.extend_from_slice(&xy);
            pinf.push(f);
        }
        let mut xy = vec![0u8; n * 64];
        let mut inf = vec![0u8; n];
        let eng = ENGINE.lock().unwrap();
        let rc = unsafe {
            ecgpu_batch_mul(eng.0, ECGPU_P256, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(),
                            inf.as_mut_ptr())
        };
        assert_eq!(rc, ECGPU_OK);
        xy.chunks(64).zip(inf).map(|(c, f)| from_wire(c, f)).collect()
    }

    /// `LinearCombination::lincomb` on the GPU (Pippenger instead of Straus; same group element).
    pub fn lincomb(terms: &[(ProjectivePoint, Scalar)]) -> ProjectivePoint {
        let n = terms.len();
        let scalars: Vec<u8> = terms.iter().flat_map(|(_, k)| k.to_repr()).collect();
        let (mut pts, mut pinf) = (Vec::with_capacity(n * 64), Vec::with_capacity(n));
        for (p, _) in terms {
            let (xy, f) = to_wire(p);
            pts.extend_from_slice(&xy);
            pinf.push(f);
        }
        let (mut xy, mut inf) = ([0u8; 64], 0u8);
        let eng = ENGINE.lock().unwrap();
        let rc = unsafe {
            ecgpu_msm(eng.0, ECGPU_P256, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), &mut inf)
        };
        assert_eq!(rc, ECGPU_OK);
        from_wire(&xy, inf)
    }

    /// The `MulBackend` plug-in.  Single-element calls keep using the CPU tables (a GPU launch for one
    /// scalar is pointless); callers with batches use the functions above.  Below `GPU_MIN_TERMS`
    /// terms `lincomb` also stays on the CPU.
    #[derive(Clone, Copy, Debug)]
    pub struct GpuBackend;
    pub const GPU_MIN_TERMS: usize = 1 << 10;

    impl MulBackend<NistP256> for GpuBackend {
        fn mul_by_generator(k: &Scalar) -> ProjectivePoint {
            <p256::arithmetic::tables::backend::PrecomputedTables as MulBackend<NistP256>>::mul_by_generator(k)
        }
        fn mul_by_generator_vartime(k: &Scalar) -> ProjectivePoint {
            <p256::arithmetic::tables::backend::PrecomputedTables as MulBackend<NistP256>>::mul_by_generator_vartime(k)
        }
        fn mul_by_generator_and_mul_add_vartime(a: &Scalar, b: &Scalar, p: &ProjectivePoint) -> ProjectivePoint {
            ProjectivePoint::lincomb_vartime(&[(ProjectivePoint::GENERATOR, *a), (*p, *b)])
        }
    }

    /// Drop-in for `ProjectivePoint::lincomb(&[(P, k)])` that moves large sums to the GPU.
    pub fn lincomb_auto(terms: &[(ProjectivePoint, Scalar)]) -> ProjectivePoint {
        if terms.len() >= GPU_MIN_TERMS {
            lincomb(terms)
        } else {
            ProjectivePoint::lincomb(terms)
        }
    }
}
// k256 does not go through MulBackend (inherent fns, k256/src/arithmetic/mul.rs:177-233); the same three
// wrappers are written against k256::{ProjectivePoint, Scalar} with ECGPU_K256 and hooked at
// `ProjectivePoint::mul_by_generator`, `Mul<Scalar>` (batch form) and `LinearCombination::lincomb`.
