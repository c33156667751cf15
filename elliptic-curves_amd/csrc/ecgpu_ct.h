// ecgpu_ct.h — kernels of the uniform-schedule entry points (ecgpu_batch_mul_ct / _mul_base_ct / _ecdh_ct; HIP only).
// The per-lane algorithms and the reference citations are in ecgpu_ctmul.h; this file adds the loads and stores, which
// follow the same rule: nothing here branches on, or computes an address from, the contents of a scalar or point record.
#pragma once

#include "ecgpu_ctmul.h"
#include "ecgpu_kernels.h"
#include "ecgpu_var.h"

namespace ecgpu {

// scalar record -> words; the range verdict (`Scalar::from_repr`, k256 scalar.rs:310-316) as a flag bit, no branch
template <class C>
__device__ __forceinline__ uint32_t ct_load_scalar(uint32_t* k, const uint8_t* scalars, size_t i) {
    load_wire<C>(k, scalars + i * WireBytes<C>::value);
    return mp_geq<C::N>(k, C::ORDER) ? CT_FLAG_BAD_SCALAR : 0u;
}

// affine record (+ identity flag) -> homogeneous coordinates, (0 : 1 : 0) for a flagged identity; the
// `AffinePoint::from_coordinates` verdict (primeorder/src/affine.rs:100-109) as a flag bit.  The record of a flagged
// identity is read and checked like any other and its verdict dropped under a mask.
template <class C>
__device__ __forceinline__ uint32_t ct_load_point(Proj<C>* p, const uint8_t* xy, const uint8_t* inf, size_t i, const Fe<C::NL>& b) {
    using F = Field<C>;
    using G = Group<C>;
    const bool ident = inf != nullptr ? inf[i] != 0 : false;       // (the pointer is an argument of the call, not data)
    uint32_t cx[C::N], cy[C::N];
    load_wire<C>(cx, xy + i * (2 * WireBytes<C>::value));
    load_wire<C>(cy, xy + i * (2 * WireBytes<C>::value) + WireBytes<C>::value);
    const bool in_range = (int)!mp_geq<C::N>(cx, C::P) & (int)!mp_geq<C::N>(cy, C::P);        // (& on purpose: no short circuit)
    Affine<C> a;
    a.x = F::from_canonical(cx).e;
    a.y = F::from_canonical(cy).e;
    const bool ok = (int)in_range & (int)G::on_curve(a, b);
    *p = ct_sel_proj<C>(ident, G::identity(), G::from_affine(a));
    return ((int)ok | (int)ident) ? 0u : CT_FLAG_BAD_POINT;
}

// out[i] = k[i] * P[i]; one lane per element, the table [P..8P] (projective) in the lane's slot of the HBM scratch that
// the variable-time kernel uses too (ecgpu_var.h: [wave][entry][row][lane], every access a coalesced 256-byte row)
template <class C>
__global__ void __launch_bounds__(BLOCK, 2)
k_var_base_ct(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points_xy, const uint8_t* __restrict__ points_inf,
              size_t n, uint32_t* __restrict__ tab, size_t tstride, uint32_t* __restrict__ proj_out, uint8_t* __restrict__ flags) {
    using G = Group<C>;
    constexpr int N = C::N;
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // < tstride
    const Fe<C::NL> b = G::curve_b();
    VarTabHbm<C> io{tab + (slot / 64) * (size_t)(8 * VarTabHbm<C>::ROWS * 64) + (slot % 64)};
    for (size_t i = slot; i < n; i += tstride) {
        uint32_t k[N];
        uint32_t f = ct_load_scalar<C>(k, scalars, i);
        Proj<C> p;
        f |= ct_load_point<C>(&p, points_xy, points_inf, i, b);
        flags[i] = (uint8_t)f;
        store_proj<C>(proj_out, i, var_base_mul_ct<C>(p, k, b, io));
    }
}

// the generator LUTs: [CT_BASE_LUTS][CT_BASE_ENTRIES][2] packed elements, read at wave-uniform addresses (every lane scans the
// same entries of the same LUT): scalar loads, the entry words arrive in scalar registers (ct_pick_words_uniform).
// Measured alternatives (profiles/r03/ct_generator_scan_variants.txt): the window's LUT staged in LDS by the workgroup and
// scanned with broadcast reads + half-slot v_cndmask (a quarter fewer vector instructions, but k256 no faster and p384 slower:
// the reads' latency is exposed at two or three waves per SIMD, and issuing them ten at a time is slower still).
template <class C>
struct CtLutHbm {
    static constexpr bool UNIFORM = true;
    const uint32_t* lut;
    __device__ void load(PackedPoint<2 * C::N>& p, int i, int entry) const {
        load_words_vec<2 * C::N>(p.w, lut + ((size_t)i * CT_BASE_ENTRIES + entry) * (2 * C::N));
    }
};

template <class C>
__global__ void __launch_bounds__(BLOCK, 2)
k_fixed_base_ct(const uint8_t* __restrict__ scalars, size_t n, const uint32_t* __restrict__ lut, uint32_t* __restrict__ proj_out,
                uint8_t* __restrict__ flags) {
    using G = Group<C>;
    constexpr int N = C::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[N];
    flags[i] = (uint8_t)ct_load_scalar<C>(k, scalars, i);
    CtLutHbm<C> t{lut};
    store_proj<C>(proj_out, i, fixed_base_mul_ct<C>(k, t, G::curve_b()));
}

// status |= OR of the n flag bytes (CT_FLAG_* are the ST_* bits); a kernel of its own so that the multiplication kernels
// have no path that depends on a verdict
static __global__ void __launch_bounds__(BLOCK) k_ct_flags(const uint8_t* __restrict__ flags, size_t n, int* status) {
    uint32_t f = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) f |= flags[i];
    if (f) atomicOr(status, (int)f);
}

}  // namespace ecgpu
