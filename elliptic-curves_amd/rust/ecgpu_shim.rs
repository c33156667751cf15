//! ecgpu_shim.rs — the reference-side binding a maintainer would add to RustCrypto/elliptic-curves (as a workspace
//! crate `ecgpu`, enabled by a cargo feature `gpu` in `k256`, `p256`, `p384`, `primeorder`).  NOT built in this repo:
//! the image has no rustc / cargo.  The raw `extern "C"` block is generated from include/ecgpu.h
//! (ecgpu_sys.rs, tools/gen_rust_sys.py) and checked against the header by tests/test_abi.py; this file is the safe
//! layer above it and the text of the patch sites.
//!
//! It exposes libecgpu.so behind the reference's own plug-in points:
//!   * `elliptic_curve::ops::LinearCombination::lincomb_vartime`
//!         primeorder/src/projective.rs:480-511 (generic curves) and k256/src/arithmetic/mul.rs:84-109 (k256)
//!   * `elliptic_curve::ops::MulByGeneratorVartime::{mul_by_generator_vartime, mul_by_generator_and_mul_add_vartime}`
//!         primeorder/src/projective.rs:923-940 -> `primeorder::MulBackend<C>` (primeorder/src/mul_backend.rs:11-40);
//!         k256: inherent fns k256/src/arithmetic/mul.rs:177-233 and the impl at :296-310
//!   * new batch entry points (a GPU needs batches; the reference's API is one element at a time):
//!         `batch_mul_by_generator_vartime`, `batch_mul_vartime`, `batch_mul_by_generator_and_mul_add_vartime`,
//!         `batch_verify_prehashed`
//!
//! SECRET SCALARS.  The fast GPU paths are VARIABLE-TIME in their scalars (zero digits are skipped, table entries and
//! buckets are addressed by scalar bits).  They stand behind the reference's `*_vartime` names ONLY
//! (`MulVartime` primeorder/src/projective.rs:888-921, `lincomb_vartime`, `mul_by_generator_vartime`): public scalars —
//! signature verification, MSMs over public data, public-key derivation from non-secret material.
//! The constant-time entry points (`Mul`, `mul_by_generator`, `diffie_hellman`: primeorder/src/projective.rs:532-557,
//! 847-886, tables/lookup.rs:43-65, k256/src/ecdh.rs:56-60) have batch forms of their own over the library's
//! uniform-schedule kernels — `batch_mul`, `batch_mul_by_generator`, `batch_diffie_hellman` below call
//! `ecgpu_batch_mul_ct` / `ecgpu_batch_mul_base_ct` / `ecgpu_batch_ecdh_ct`: the reference's constant-time algorithm
//! itself (fixed digit count, every table entry read and one kept under a mask, complete formulas), with no branch and
//! no address computed from scalar or point data (include/ecgpu.h, "uniform-schedule variants"; tools/ct_isa_check.py).
//! The constant-time `lincomb` (:484-496; k256 mul.rs:84-98) goes to `ecgpu_lincomb_ct` (`lincomb` below): one
//! uniform-schedule multiplication per term and a tree of complete additions — never to the bucket method, which is
//! variable-time by construction.
//! Memory hygiene on these paths: the wire copies of secret scalars and of shared secrets live in `Zeroizing` buffers here,
//! and the library zeroes its own device-side staging and scratch after every `_ct` call (include/ecgpu.h, ecgpu_wipe) —
//! the counterpart of the reference's zeroize-on-drop `NonZeroScalar` / `SharedSecret`.
//!
//! Build: `links = "ecgpu"` + a build.rs emitting `cargo:rustc-link-lib=dylib=ecgpu` and
//! `cargo:rustc-link-search=<repo>/elliptic-curves_amd/lib`.

#[path = "ecgpu_sys.rs"]
pub mod sys;

use core::ffi::c_int;
use std::sync::{LazyLock, Mutex};

use elliptic_curve::{
    point::AffineCoordinates,
    sec1::{FromSec1Point, ModulusSize, Sec1Point},
    CurveArithmetic, FieldBytes, PrimeField,
};
use sys::*;
use zeroize::Zeroizing;

/// One GPU: the analogue of `static BASEPOINT_TABLE: LazyLock<..>` (k256/src/arithmetic/tables.rs:18,
/// primeorder/src/tables/basepoint.rs:29-31) — created on first use, holds the device-resident comb tables.
pub struct Engine(*mut EcgpuCtx);
unsafe impl Send for Engine {}

pub static ENGINE: LazyLock<Option<Mutex<Engine>>> = LazyLock::new(|| {
    let mut ctx = core::ptr::null_mut();
    // no gfx950 device -> None: every adapter below then reports "not handled" and the CPU code runs
    if unsafe { ecgpu_init(&mut ctx, 0) } != ECGPU_OK {
        return None;
    }
    // The generator tables follow the library's default footprint policy (ECGPU_TABLE_ADAPTIVE: 34 MB until the device has
    // multiplied 2^26 scalars by G, 21.5 GB only from 2^29 on — the reference's own table is 30-60 KB and lazily built,
    // primeorder/src/tables/basepoint.rs:29-76).  Until a tier is reached the rate is the narrower table's — per 2^20 k256 scalars
    // about 0.93 ms at 16 bits and 0.74 ms at 22 against 0.59 ms at 26 (bench.py prints all three: `fixed_k256_tier_ms`; its
    // headline runs on the widest, policy eager) — and the ONE call that crosses a tier builds the next table before it returns
    // (1 / 6 / 55 ms on the calling thread, also on an asynchronous context).  A service that wants the steady-state rate from
    // its first batch sets ECGPU_TABLE_POLICY=eager; ECGPU_TABLE_BUDGET_MB caps a table whatever the policy.
    if std::env::var("ECGPU_TABLE_POLICY").map_or(false, |v| v == "eager") {
        unsafe { ecgpu_set_table_policy(ctx, ECGPU_TABLE_EAGER) };
    }
    if let Some(mb) = std::env::var("ECGPU_TABLE_BUDGET_MB").ok().and_then(|v| v.parse::<usize>().ok()) {
        unsafe { ecgpu_set_table_budget(ctx, mb << 20) };
    }
    Some(Mutex::new(Engine(ctx)))
});

/// All GPUs of the node: used for MSMs of `NODE_MIN_TERMS` terms and more.
pub struct Node(*mut EcgpuGroup);
unsafe impl Send for Node {}

pub static NODE: LazyLock<Option<Mutex<Node>>> = LazyLock::new(|| {
    // every gfx950 device the runtime shows (ecgpu_device_count), capped by ECGPU_DEVICES; fewer than two: no group, the
    // single-GPU engine handles everything
    let have = unsafe { ecgpu_device_count() }.max(0) as usize;
    let ndev = std::env::var("ECGPU_DEVICES").ok().and_then(|v| v.parse::<usize>().ok()).map_or(have, |v| v.min(have));
    if ndev < 2 {
        return None;
    }
    let devices: Vec<c_int> = (0..ndev as c_int).collect();
    let mut g = core::ptr::null_mut();
    if unsafe { ecgpu_group_init(&mut g, devices.as_ptr(), devices.len() as c_int) } != ECGPU_OK {
        return None;
    }
    // `lincomb` must return: the exchange of the partial sums is given up after this long (a collective that hangs is the failure
    // RCCL has shown) and the call completes over peer copies — include/ecgpu.h ecgpu_group_set_exchange_timeout, default 10 s
    if let Some(sec) = std::env::var("ECGPU_EXCHANGE_TIMEOUT_S").ok().and_then(|v| v.parse::<f64>().ok()) {
        unsafe { ecgpu_group_set_exchange_timeout(g, sec) };
    }
    Some(Mutex::new(Node(g)))
});

/// Below this many terms a sum stays on the CPU (a launch costs ~1 ms end to end).
pub const GPU_MIN_TERMS: usize = 1 << 10;
/// The same threshold for the constant-time `lincomb` (n uniform-schedule multiplications, ~20 ns each at full occupancy).
pub const GPU_MIN_TERMS_CT: usize = 1 << 8;
/// From this many terms on an MSM is spread over all GPUs of the node.
pub const NODE_MIN_TERMS: usize = 1 << 22;

/// What the adapters need to know about a curve of the reference.
pub trait GpuCurve: CurveArithmetic
where
    Self::FieldBytesSize: ModulusSize,
{
    /// `ECGPU_K256` ... (include/ecgpu.h)
    const ID: c_int;
}
impl GpuCurve for k256::Secp256k1 {
    const ID: c_int = ECGPU_K256;
}
impl GpuCurve for p256::NistP256 {
    const ID: c_int = ECGPU_P256;
}
impl GpuCurve for p384::NistP384 {
    const ID: c_int = ECGPU_P384;
}
// p224 / p192 / p521 / sm2 / bp256 / bp384: the same one-liner with their ids.

type Proj<C> = <C as CurveArithmetic>::ProjectivePoint;
type Aff<C> = <C as CurveArithmetic>::AffinePoint;
type Sc<C> = <C as CurveArithmetic>::Scalar;

fn field_len<C: GpuCurve>() -> usize
where
    C::FieldBytesSize: ModulusSize,
{
    unsafe { ecgpu_field_bytes(C::ID) }
}

/// Wire format of include/ecgpu.h: scalars = `Scalar::to_repr()`, points = affine x || y + identity flag.
fn scalars_to_wire<C: GpuCurve>(ks: impl Iterator<Item = Sc<C>>) -> Vec<u8>
where
    C::FieldBytesSize: ModulusSize,
{
    ks.flat_map(|k| k.to_repr().as_ref().to_vec()).collect()
}

/// The same for SECRET scalars: the wire copy is wiped when it goes out of scope.
fn secret_scalars_to_wire<C: GpuCurve>(ks: impl Iterator<Item = Sc<C>>) -> Zeroizing<Vec<u8>>
where
    C::FieldBytesSize: ModulusSize,
{
    let mut out = Zeroizing::new(Vec::new());
    for k in ks {
        let repr = Zeroizing::new(k.to_repr());
        out.extend_from_slice(repr.as_ref());
    }
    out
}

fn points_to_wire<C: GpuCurve>(ps: impl Iterator<Item = Proj<C>>) -> (Vec<u8>, Vec<u8>)
where
    C::FieldBytesSize: ModulusSize,
    Aff<C>: AffineCoordinates<FieldRepr = FieldBytes<C>>,
{
    // to_affine per point (one inversion each on the CPU); a caller holding many projective points ships X || Y || Z
    // to ecgpu_batch_normalize instead (BatchNormalize::batch_normalize, primeorder/src/projective.rs:452-478)
    let l = field_len::<C>();
    let (mut xy, mut inf) = (Vec::new(), Vec::new());
    for p in ps {
        let a: Aff<C> = p.into();
        let ident = bool::from(elliptic_curve::group::Group::is_identity(&p));
        if ident {
            xy.extend(core::iter::repeat(0u8).take(2 * l));
        } else {
            xy.extend_from_slice(a.x().as_ref());
            xy.extend_from_slice(a.y().as_ref());
        }
        inf.push(ident as u8);
    }
    (xy, inf)
}

fn point_from_wire<C: GpuCurve>(xy: &[u8], inf: u8) -> Proj<C>
where
    C::FieldBytesSize: ModulusSize,
    Aff<C>: FromSec1Point<C>,
{
    if inf != 0 {
        return <Proj<C> as elliptic_curve::group::Group>::identity();
    }
    let l = xy.len() / 2;
    let x = FieldBytes::<C>::try_from(&xy[..l]).expect("field length");
    let y = FieldBytes::<C>::try_from(&xy[l..]).expect("field length");
    // the device only returns points of the curve: from_sec1_point cannot fail here
    let a: Aff<C> = Option::from(Aff::<C>::from_sec1_point(&Sec1Point::<C>::from_affine_coordinates(&x, &y, false))).expect("on curve");
    a.into()
}

/// The safe batch layer.  Every function returns `None` when there is no usable GPU — no device, or a runtime failure of
/// this call (ECGPU_ERR_HIP, ECGPU_ERR_OOM, ECGPU_ERR_NO_DEVICE) — and the caller then runs the reference's CPU code; it
/// panics only on a contract violation (ECGPU_ERR_ARG / ECGPU_ERR_CURVE, or a range / on-curve error for values that came
/// out of the reference's own validated types, which cannot be out of range).
pub mod gpu {
    use super::*;

    /// `Some(())` on success, `None` for "the GPU could not do it this time" (the CPU path takes over).
    fn check(rc: c_int) -> Option<()> {
        match rc {
            ECGPU_OK => Some(()),
            ECGPU_ERR_HIP | ECGPU_ERR_OOM | ECGPU_ERR_NO_DEVICE => None,
            _ => panic!("libecgpu: contract violation, error {rc}"),
        }
    }

    /// `k[i] * G` — batch form of `MulByGeneratorVartime::mul_by_generator_vartime`.
    pub fn batch_mul_by_generator_vartime<C: GpuCurve>(ks: &[Sc<C>]) -> Option<Vec<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let l = field_len::<C>();
        let scalars = scalars_to_wire::<C>(ks.iter().copied());
        let (mut xy, mut inf) = (vec![0u8; ks.len() * 2 * l], vec![0u8; ks.len()]);
        check(unsafe { ecgpu_batch_mul_base(eng.0, C::ID, scalars.as_ptr(), ks.len(), xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        Some(xy.chunks(2 * l).zip(inf).map(|(c, f)| point_from_wire::<C>(c, f)).collect())
    }

    /// `k[i] * P[i]` — batch form of `MulVartime::mul_vartime` (primeorder/src/projective.rs:888-921).
    pub fn batch_mul_vartime<C: GpuCurve>(terms: &[(Proj<C>, Sc<C>)]) -> Option<Vec<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (terms.len(), field_len::<C>());
        let scalars = scalars_to_wire::<C>(terms.iter().map(|t| t.1));
        let (pts, pinf) = points_to_wire::<C>(terms.iter().map(|t| t.0));
        let (mut xy, mut inf) = (vec![0u8; n * 2 * l], vec![0u8; n]);
        check(unsafe {
            ecgpu_batch_mul(eng.0, C::ID, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), inf.as_mut_ptr())
        })?;
        Some(xy.chunks(2 * l).zip(inf).map(|(c, f)| point_from_wire::<C>(c, f)).collect())
    }

    /// `k[i] * G` — batch form of the constant-time `mul_by_generator` (k256/src/arithmetic/mul.rs:180-197,
    /// primeorder/src/tables/basepoint.rs:82-99) on the uniform-schedule kernel.
    pub fn batch_mul_by_generator<C: GpuCurve>(ks: &[Sc<C>]) -> Option<Vec<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let l = field_len::<C>();
        let scalars = secret_scalars_to_wire::<C>(ks.iter().copied());
        let (mut xy, mut inf) = (vec![0u8; ks.len() * 2 * l], vec![0u8; ks.len()]);
        check(unsafe { ecgpu_batch_mul_base_ct(eng.0, C::ID, scalars.as_ptr(), ks.len(), xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        Some(xy.chunks(2 * l).zip(inf).map(|(c, f)| point_from_wire::<C>(c, f)).collect())
    }

    /// `k[i] * P[i]` — batch form of the constant-time `impl Mul<Scalar> for ProjectivePoint`
    /// (primeorder/src/projective.rs:847-886, k256/src/arithmetic/mul.rs:249-274) on the uniform-schedule kernel.
    pub fn batch_mul<C: GpuCurve>(terms: &[(Proj<C>, Sc<C>)]) -> Option<Vec<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (terms.len(), field_len::<C>());
        let scalars = secret_scalars_to_wire::<C>(terms.iter().map(|t| t.1));
        let (pts, pinf) = points_to_wire::<C>(terms.iter().map(|t| t.0));
        let (mut xy, mut inf) = (Zeroizing::new(vec![0u8; n * 2 * l]), vec![0u8; n]);    // k P for a secret k: wiped on drop
        check(unsafe {
            ecgpu_batch_mul_ct(eng.0, C::ID, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), inf.as_mut_ptr())
        })?;
        Some(xy.chunks(2 * l).zip(inf).map(|(c, f)| point_from_wire::<C>(c, f)).collect())
    }

    /// `diffie_hellman(secret[i], public[i]).raw_secret_bytes()` (k256/src/ecdh.rs:56-60, p256/src/ecdh.rs; the function
    /// is `elliptic_curve::ecdh::diffie_hellman`, (public * secret).to_affine().x()) — the x-coordinates, `None` in a slot
    /// whose product is the identity (cannot happen for a `NonZeroScalar` and a `PublicKey`).
    /// The x-coordinates come back in zeroize-on-drop wrappers (`SharedSecret` is one, elliptic-curve ecdh.rs): the caller
    /// builds `SharedSecret::from(bytes)` from each and drops the vector.
    pub fn batch_diffie_hellman<C: GpuCurve>(pairs: &[(Sc<C>, Aff<C>)]) -> Option<Vec<Option<Zeroizing<FieldBytes<C>>>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (pairs.len(), field_len::<C>());
        let scalars = secret_scalars_to_wire::<C>(pairs.iter().map(|t| t.0));
        let (pts, _) = points_to_wire::<C>(pairs.iter().map(|t| Proj::<C>::from(t.1)));
        let (mut x, mut ok) = (Zeroizing::new(vec![0u8; n * l]), vec![0u8; n]);
        check(unsafe { ecgpu_batch_ecdh_ct(eng.0, C::ID, scalars.as_ptr(), pts.as_ptr(), n, x.as_mut_ptr(), ok.as_mut_ptr()) })?;
        Some(x.chunks(l).zip(ok).map(|(c, f)| (f != 0).then(|| Zeroizing::new(FieldBytes::<C>::try_from(c).expect("field bytes")))).collect())
    }

    /// `sum_i k[i] * P[i]` — `LinearCombination::lincomb`, the CONSTANT-TIME form (primeorder/src/projective.rs:484-496,
    /// k256/src/arithmetic/mul.rs:84-98): `ecgpu_lincomb_ct`, one uniform-schedule multiplication per term and a tree of
    /// complete additions.  Costs n constant-time multiplications (no bucket method): worth it from a few hundred terms
    /// (`GPU_MIN_TERMS_CT`), where the reference's Straus loop over n per-term tables has left the CPU caches.
    pub fn lincomb<C: GpuCurve>(terms: &[(Proj<C>, Sc<C>)]) -> Option<Proj<C>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (terms.len(), field_len::<C>());
        let scalars = secret_scalars_to_wire::<C>(terms.iter().map(|t| t.1));
        let (pts, pinf) = points_to_wire::<C>(terms.iter().map(|t| t.0));
        let (mut xy, mut inf) = (Zeroizing::new(vec![0u8; 2 * l]), 0u8);
        check(unsafe { ecgpu_lincomb_ct(eng.0, C::ID, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), &mut inf) })?;
        Some(point_from_wire::<C>(&xy, inf))
    }

    /// `sum_i k[i] * P[i]` over SEC1-COMPRESSED public keys (33-byte `tag || x` encodings as they arrive on the wire:
    /// `PublicKey::from_sec1_bytes`, `FromSec1Point` -> `DecompressPoint::decompress`, primeorder/src/affine.rs:183-200,
    /// k256/src/arithmetic/affine.rs:261-280): the square roots run on the device (`ecgpu_msm_compressed`).  `None` in the
    /// outer `Option` = no GPU; `Some(None)` = some encoding is not a point of the curve (the reference's `CtOption::None`).
    pub fn lincomb_vartime_sec1<C: GpuCurve>(keys: &[&[u8]], ks: &[Sc<C>]) -> Option<Option<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (keys.len(), field_len::<C>());
        assert!(ks.len() == n);
        let (mut xs, mut tags) = (vec![0u8; n * l], vec![0u8; n]);
        for (i, k) in keys.iter().enumerate() {
            match k.len() {
                1 if k[0] == 0 => {}                                            // the identity: tag 0, x = 0
                m if m == l + 1 => {
                    tags[i] = k[0];
                    xs[i * l..(i + 1) * l].copy_from_slice(&k[1..]);
                }
                _ => return Some(None),                                         // uncompressed / hybrid forms: the caller's CPU path
            }
        }
        let scalars = scalars_to_wire::<C>(ks.iter().copied());
        let (mut xy, mut inf) = (vec![0u8; 2 * l], 0u8);
        match unsafe { ecgpu_msm_compressed(eng.0, C::ID, scalars.as_ptr(), xs.as_ptr(), tags.as_ptr(), n, xy.as_mut_ptr(), &mut inf) } {
            ECGPU_ERR_POINT => Some(None),
            rc => check(rc).map(|_| Some(point_from_wire::<C>(&xy, inf))),
        }
    }

    /// `sum_i k[i] * P[i]` — `LinearCombination::lincomb_vartime` (Pippenger instead of Straus; same group element).
    /// One GPU up to NODE_MIN_TERMS terms, all GPUs of the node beyond (ecgpu_group_msm: term shards, one exchange of
    /// per-window partial sums over xGMI, one combining step).
    pub fn lincomb_vartime<C: GpuCurve>(terms: &[(Proj<C>, Sc<C>)]) -> Option<Proj<C>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let (n, l) = (terms.len(), field_len::<C>());
        let scalars = scalars_to_wire::<C>(terms.iter().map(|t| t.1));
        let (pts, pinf) = points_to_wire::<C>(terms.iter().map(|t| t.0));
        let (mut xy, mut inf) = (vec![0u8; 2 * l], 0u8);
        if n >= NODE_MIN_TERMS {
            if let Some(node) = NODE.as_ref().and_then(|m| m.lock().ok()) {
                // a failed group call (one GPU of the node gone, out of memory) falls through to the single-GPU engine
                if check(unsafe {
                    ecgpu_group_msm(node.0, C::ID, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), &mut inf)
                })
                .is_some()
                {
                    return Some(point_from_wire::<C>(&xy, inf));
                }
            }
        }
        let eng = ENGINE.as_ref()?.lock().ok()?;
        check(unsafe { ecgpu_msm(eng.0, C::ID, scalars.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(), &mut inf) })?;
        Some(point_from_wire::<C>(&xy, inf))
    }

    /// `a[i] * G + b[i] * P[i]` — batch form of `mul_by_generator_and_mul_add_vartime`
    /// (primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310).
    pub fn batch_mul_by_generator_and_mul_add_vartime<C: GpuCurve>(abp: &[(Sc<C>, Sc<C>, Proj<C>)]) -> Option<Vec<Proj<C>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C> + AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let (n, l) = (abp.len(), field_len::<C>());
        let a = scalars_to_wire::<C>(abp.iter().map(|t| t.0));
        let b = scalars_to_wire::<C>(abp.iter().map(|t| t.1));
        let (pts, pinf) = points_to_wire::<C>(abp.iter().map(|t| t.2));
        let (mut xy, mut inf) = (vec![0u8; n * 2 * l], vec![0u8; n]);
        check(unsafe {
            ecgpu_batch_mul_base_and_mul_add(eng.0, C::ID, a.as_ptr(), b.as_ptr(), pts.as_ptr(), pinf.as_ptr(), n, xy.as_mut_ptr(),
                                             inf.as_mut_ptr())
        })?;
        Some(xy.chunks(2 * l).zip(inf).map(|(c, f)| point_from_wire::<C>(c, f)).collect())
    }

    /// Batch form of `ecdsa::hazmat::verify_prehashed` (ecdsa 0.17.0; p256/src/ecdsa.rs:69, k256/src/ecdsa.rs:99-106):
    /// `z` = `bits2field(digest)`, `(r, s)` the signature scalars as bytes, `q` the verifying keys.  One verdict per
    /// element; a bad element never fails the batch.  `normalize_s` = `C::NORMALIZE_S` (true for k256).
    pub fn batch_verify_prehashed<C: GpuCurve>(z: &[FieldBytes<C>], r: &[FieldBytes<C>], s: &[FieldBytes<C>], q: &[Aff<C>],
                                               normalize_s: bool) -> Option<Vec<bool>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: AffineCoordinates<FieldRepr = FieldBytes<C>>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let n = z.len();
        assert!(r.len() == n && s.len() == n && q.len() == n);
        let cat = |v: &[FieldBytes<C>]| v.iter().flat_map(|b| b.as_ref().to_vec()).collect::<Vec<u8>>();
        let qxy: Vec<u8> = q.iter().flat_map(|p| [p.x().as_ref(), p.y().as_ref()].concat()).collect();
        let mut ok = vec![0u8; n];
        check(unsafe {
            ecgpu_ecdsa_verify_batch(eng.0, C::ID, cat(z).as_ptr(), cat(r).as_ptr(), cat(s).as_ptr(), qxy.as_ptr(), n,
                                     normalize_s as c_int, ok.as_mut_ptr())
        })?;
        Some(ok.into_iter().map(|b| b != 0).collect())
    }

    /// Batch form of `sm2::dsa::VerifyingKey::new(distid, pk)?.verify(msg, &sig)` for signers that share one distinguishing
    /// identifier and messages of one length (sm2/src/distid.rs:21-44 `hash_z`, sm2/src/dsa/verifying.rs:126-171): the two SM3
    /// hashes run on the device.  `keys` = affine x || y (64 bytes each), `sigs` = r || s (64 bytes each).
    pub fn sm2dsa_batch_verify(distid: &[u8], keys: &[[u8; 64]], msgs: &[u8], msg_len: usize, sigs: &[[u8; 64]]) -> Option<Vec<bool>> {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let n = keys.len();
        assert!(sigs.len() == n && msgs.len() == n * msg_len && distid.len() <= 8191);
        let mut ok = vec![0u8; n];
        check(unsafe {
            ecgpu_sm2dsa_verify_msg_batch(eng.0, distid.as_ptr(), distid.len(), keys.as_ptr() as *const u8, msgs.as_ptr(), msg_len,
                                          sigs.as_ptr() as *const u8, n, ok.as_mut_ptr())
        })?;
        Some(ok.into_iter().map(|b| b != 0).collect())
    }

    /// Batch form of `bignp256::ecdsa::VerifyingKey::from_bytes(pk)?.verify(msg, &sig)` for messages of one length
    /// (bignp256/src/ecdsa/verifying.rs:100-169): both belt-hash computations run on the device.  `keys` = affine x || y
    /// (64 bytes each, little-endian like `FIELD_ENDIANNESS`), `sigs` = `Signature::to_bytes()` (48 bytes each).
    pub fn bign_batch_verify(keys: &[[u8; 64]], msgs: &[u8], msg_len: usize, sigs: &[[u8; 48]]) -> Option<Vec<bool>> {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let n = keys.len();
        assert!(sigs.len() == n && msgs.len() == n * msg_len);
        let mut ok = vec![0u8; n];
        check(unsafe {
            ecgpu_bign_verify_msg_batch(eng.0, keys.as_ptr() as *const u8, msgs.as_ptr(), msg_len, sigs.as_ptr() as *const u8, n,
                                        ok.as_mut_ptr())
        })?;
        Some(ok.into_iter().map(|b| b != 0).collect())
    }

    /// Batch form of `VerifyingKey::<C>::recover_from_prehash` (ecdsa 0.17.0 recovery.rs; the reference's vectors:
    /// k256/src/ecdsa.rs:190-262): `recovery_id[i]` = `RecoveryId::to_byte()`.  `None` per element where the reference
    /// returns `Err` (id does not parse, candidate x >= p or off the curve, identity key, high s under NORMALIZE_S).
    pub fn batch_recover_from_prehash<C: GpuCurve>(z: &[FieldBytes<C>], r: &[FieldBytes<C>], s: &[FieldBytes<C>], recovery_id: &[u8],
                                                   normalize_s: bool) -> Option<Vec<Option<Aff<C>>>>
    where
        C::FieldBytesSize: ModulusSize,
        Aff<C>: FromSec1Point<C>,
    {
        let eng = ENGINE.as_ref()?.lock().ok()?;
        let n = z.len();
        assert!(r.len() == n && s.len() == n && recovery_id.len() == n);
        let l = field_len::<C>();
        let cat = |v: &[FieldBytes<C>]| v.iter().flat_map(|b| b.as_ref().to_vec()).collect::<Vec<u8>>();
        let (mut xy, mut ok) = (vec![0u8; n * 2 * l], vec![0u8; n]);
        check(unsafe {
            ecgpu_ecdsa_recover_batch(eng.0, C::ID, cat(z).as_ptr(), cat(r).as_ptr(), cat(s).as_ptr(), recovery_id.as_ptr(), n,
                                      normalize_s as c_int, xy.as_mut_ptr(), ok.as_mut_ptr())
        })?;
        Some(xy.chunks(2 * l).zip(ok).map(|(c, f)| if f != 0 { Some(elliptic_curve::group::Curve::to_affine(&point_from_wire::<C>(c, 0))) } else { None }).collect())
    }
}

// =====================================================================================================================
// Patch sites in the reference (what `--features gpu` adds; `ecgpu` = this crate)
// =====================================================================================================================
//
// (1) primeorder/src/projective.rs:498-510 — `impl<C> LinearCombination<[(Self, Scalar<C>)]> for ProjectivePoint<C>`,
//     at the top of `fn lincomb_vartime` (p256, p384, p224, p521, sm2, bp256, bp384: every primeorder curve):
//
//         #[cfg(feature = "gpu")]
//         if points_and_scalars.len() >= ecgpu::GPU_MIN_TERMS {
//             if let Some(sum) = ecgpu::gpu::lincomb_vartime::<C>(points_and_scalars) {
//                 return sum;
//             }
//         }
//
//     and at the top of `fn lincomb` (:484-496, the CONSTANT-TIME form), the uniform-schedule entry point — never the
//     bucket method:
//
//         #[cfg(feature = "gpu")]
//         if points_and_scalars.len() >= ecgpu::GPU_MIN_TERMS_CT {
//             if let Some(sum) = ecgpu::gpu::lincomb::<C>(points_and_scalars) {
//                 return sum;
//             }
//         }
//
// (1b) primeorder/src/projective.rs:847-886 (`impl Mul<Scalar<C>> for ProjectivePoint<C>`) and k256/src/arithmetic/mul.rs:249-274
//     are single-element operators and stay on the CPU; their constant-time BATCH forms are new inherent functions beside
//     them, over the uniform-schedule kernels:
//
//         #[cfg(feature = "gpu")]
//         impl<C: PrimeCurveParams + ecgpu::GpuCurve> ProjectivePoint<C> {
//             /// `terms[i].1 * terms[i].0` for a whole slice, constant time in the scalars (GPU, uniform schedule).
//             pub fn batch_mul(terms: &[(Self, Scalar<C>)]) -> Vec<Self> {
//                 ecgpu::gpu::batch_mul::<C>(terms).unwrap_or_else(|| terms.iter().map(|(p, k)| *p * *k).collect())
//             }
//             /// `ks[i] * G` for a whole slice, constant time.
//             pub fn batch_mul_by_generator(ks: &[Scalar<C>]) -> Vec<Self> {
//                 ecgpu::gpu::batch_mul_by_generator::<C>(ks).unwrap_or_else(|| ks.iter().map(Self::mul_by_generator).collect())
//             }
//         }
//
//     and {k256,p256,p384}/src/ecdh.rs gain `pub fn diffie_hellman_batch(pairs: &[(NonZeroScalar, PublicKey)]) ->
//     Vec<SharedSecret>` on `ecgpu::gpu::batch_diffie_hellman` (fallback: `elliptic_curve::ecdh::diffie_hellman` per pair).
//
// (2) k256/src/arithmetic/mul.rs:100-108 — the same three lines at the top of k256's own
//     `LinearCombination<[(ProjectivePoint, Scalar)]>::lincomb_vartime` (k256 does not use primeorder), with
//     `ecgpu::gpu::lincomb_vartime::<Secp256k1>`; the array form at :75-82 forwards to the slice form above
//     GPU_MIN_TERMS.  k256's constant-time `fn lincomb` (:84-98) gets the `ecgpu::gpu::lincomb::<Secp256k1>` lines of (1).
//
// (3) k256/src/arithmetic/mul.rs:205-232 (`mul_by_generator_vartime`) and :303-310
//     (`mul_by_generator_and_mul_add_vartime`) are single-element calls: they stay on the CPU (a launch for one scalar
//     is pointless).  Their batch forms are new inherent functions next to them:
//
//         #[cfg(feature = "gpu")]
//         impl ProjectivePoint {
//             /// `ks[i] * G` for a whole slice (GPU; variable time).
//             pub fn batch_mul_by_generator_vartime(ks: &[Scalar]) -> Vec<ProjectivePoint> {
//                 ecgpu::gpu::batch_mul_by_generator_vartime::<Secp256k1>(ks)
//                     .unwrap_or_else(|| ks.iter().map(ProjectivePoint::mul_by_generator_vartime).collect())
//             }
//             /// `terms[i].1 * terms[i].0` for a whole slice (GPU; variable time).
//             pub fn batch_mul_vartime(terms: &[(ProjectivePoint, Scalar)]) -> Vec<ProjectivePoint> {
//                 ecgpu::gpu::batch_mul_vartime::<Secp256k1>(terms)
//                     .unwrap_or_else(|| terms.iter().map(|(p, k)| p.mul_vartime(k)).collect())
//             }
//         }
//
//     and `k256/src/schnorr/verifying.rs:76-99` / `k256/src/ecdsa.rs` gain `verify_batch` functions built on
//     `ecgpu::gpu::batch_verify_prehashed` resp. `ecgpu_schnorr_verify_raw_batch`.
//
// (4) primeorder curves select their generator-multiplication backend through `PrimeCurveParams::Backend`
//     (primeorder/src/lib.rs:62; p256/src/arithmetic/tables.rs:24-44).  `GpuBackend` below is such a backend: single calls
//     delegate to the curve's CPU tables, and it is the type the batch functions hang off for primeorder curves.

/// `MulBackend` plug-in for the primeorder curves (primeorder/src/mul_backend.rs:11-40).
#[derive(Clone, Copy, Debug)]
pub struct GpuBackend<Cpu>(core::marker::PhantomData<Cpu>);

impl<C, Cpu> primeorder::MulBackend<C> for GpuBackend<Cpu>
where
    C: primeorder::PrimeCurveParams + GpuCurve,
    C::FieldBytesSize: ModulusSize,
    Cpu: primeorder::MulBackend<C>,
{
    // one scalar: the CPU tables (constant time, as the trait promises)
    fn mul_by_generator(k: &primeorder::Scalar<C>) -> primeorder::ProjectivePoint<C> {
        Cpu::mul_by_generator(k)
    }
    fn mul_by_generator_vartime(k: &primeorder::Scalar<C>) -> primeorder::ProjectivePoint<C> {
        Cpu::mul_by_generator_vartime(k)
    }
    fn mul_by_generator_and_mul_add_vartime(a: &primeorder::Scalar<C>, b: &primeorder::Scalar<C>,
                                            p: &primeorder::ProjectivePoint<C>) -> primeorder::ProjectivePoint<C> {
        Cpu::mul_by_generator_and_mul_add_vartime(a, b, p)
    }
}
// p256/src/arithmetic/tables.rs:44 under `--features gpu`:
//     pub type Backend = ecgpu::GpuBackend<backend::PrecomputedTables>;
// and the batch functions for p256 callers:
//     ecgpu::gpu::batch_mul_by_generator_vartime::<NistP256>(&ks), ecgpu::gpu::batch_mul_vartime::<NistP256>(&terms), ...
