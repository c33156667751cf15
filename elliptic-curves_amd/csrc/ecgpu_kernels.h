// ecgpu_kernels.h — gfx950 kernels of the batch scalar-mul engine (HIP only).
//
// Data layout in HBM
//   scalars / points in, affine points out : the wire format of include/ecgpu.h (big-endian
//       records of L resp. 2L bytes).  One lane owns one record; a wave touches 64 consecutive
//       records = one contiguous 2 KiB / 4 KiB span, loaded and stored as 16-byte vectors and
//       byte-swapped in registers.
//   projective scratch  [n][3][N] u32, internal field form (weak residues / Montgomery)
//   basepoint table     [nwin][2^(W-1)][2][N] u32 affine, internal form: entry (j, e) = e*2^(Wj)*G
//   variable-base table [8][3][N][T] u32: multiples 1..8 of each thread's point, thread-minor so
//       that a wave's accesses to one limb are contiguous.
#pragma once

#include <hip/hip_runtime.h>

#include "ecgpu_point.h"
#include "ecgpu_recode.h"

namespace ecgpu {

enum : int { ST_BAD_SCALAR = 1, ST_BAD_POINT = 2 };

constexpr int BLOCK = 256;

// ---- 16-byte vector access helpers ---------------------------------------------------------------

template <int N>
__device__ __forceinline__ void load_limbs_vec(uint32_t* dst, const uint32_t* src) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        uint4 v = s[i];
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
}
template <int N>
__device__ __forceinline__ void store_limbs_vec(uint32_t* dst, const uint32_t* src) {
    uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int i = 0; i < N / 4; i++) d[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}
// big-endian record of N words -> little-endian limbs
template <int N>
__device__ __forceinline__ void load_be_vec(uint32_t* limbs, const uint8_t* bytes) {
    uint32_t w[N];
    load_limbs_vec<N>(w, reinterpret_cast<const uint32_t*>(bytes));
#pragma unroll
    for (int i = 0; i < N; i++) limbs[i] = bswap32(w[N - 1 - i]);
}
template <int N>
__device__ __forceinline__ void store_be_vec(uint8_t* bytes, const uint32_t* limbs) {
    uint32_t w[N];
#pragma unroll
    for (int i = 0; i < N; i++) w[N - 1 - i] = bswap32(limbs[i]);
    store_limbs_vec<N>(reinterpret_cast<uint32_t*>(bytes), w);
}

template <class C>
__device__ __forceinline__ void store_proj(uint32_t* base, size_t idx, const Proj<C>& p) {
    uint32_t* d = base + idx * (3 * C::N);
    store_limbs_vec<C::N>(d, p.x.v);
    store_limbs_vec<C::N>(d + C::N, p.y.v);
    store_limbs_vec<C::N>(d + 2 * C::N, p.z.v);
}
template <class C>
__device__ __forceinline__ Proj<C> load_proj(const uint32_t* base, size_t idx) {
    Proj<C> p;
    const uint32_t* s = base + idx * (3 * C::N);
    load_limbs_vec<C::N>(p.x.v, s);
    load_limbs_vec<C::N>(p.y.v, s + C::N);
    load_limbs_vec<C::N>(p.z.v, s + 2 * C::N);
    return p;
}

// scalar record -> limbs, flags out-of-range scalars (Scalar::from_repr, k256 scalar.rs:310-316)
template <class C>
__device__ __forceinline__ void load_scalar(uint32_t* k, const uint8_t* scalars, size_t i, int* status) {
    load_be_vec<C::N>(k, scalars + i * (4 * C::N));
    if (mp_geq<C::N>(k, C::ORDER)) atomicOr(status, ST_BAD_SCALAR);
}

// affine point record -> internal form; returns false for the identity.  Flags coordinates >= p
// and off-curve points (AffinePoint::from_coordinates, primeorder/src/affine.rs:100-109).
template <class C>
__device__ __forceinline__ bool load_affine(Affine<C>* a, const uint8_t* xy, const uint8_t* inf, size_t i,
                                            const Fe<C::N>& b, int* status) {
    using F = Field<C>;
    if (inf != nullptr && inf[i]) return false;
    Fe<C::N> cx, cy;
    load_be_vec<C::N>(cx.v, xy + i * (8 * C::N));
    load_be_vec<C::N>(cy.v, xy + i * (8 * C::N) + 4 * C::N);
    bool ok = !mp_geq<C::N>(cx.v, C::P) && !mp_geq<C::N>(cy.v, C::P);
    a->x = F::from_canonical(cx);
    a->y = F::from_canonical(cy);
    ok = ok && Group<C>::on_curve(*a, b);
    if (!ok) atomicOr(status, ST_BAD_POINT);
    return true;
}

// ---- basepoint table construction -------------------------------------------------------------------
// Replaces the lazily built `BasepointTable` (primeorder/src/tables/basepoint.rs:41-76, k256
// tables.rs:11-18: 33/49 LUTs of 8 projective multiples) by one signed-window comb table of affine
// entries sized for HBM/L2 instead of a CPU L1.

// bases[j] = 2^(W*j) * G, j < nwin (one thread; nwin*W doublings)
template <class C>
__global__ void k_window_bases(uint32_t* bases, int w, int nwin) {
    using G = Group<C>;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Affine<C> g;
#pragma unroll
    for (int i = 0; i < C::N; i++) { g.x.v[i] = C::GX[i]; g.y.v[i] = C::GY[i]; }
    g.x = Field<C>::from_canonical(g.x);
    g.y = Field<C>::from_canonical(g.y);
    Fe<C::N> b = G::curve_b();
    Proj<C> p = G::from_affine(g);
    for (int j = 0; j < nwin; j++) {
        store_proj<C>(bases, j, p);
        for (int s = 0; s < w; s++) p = G::dbl(p, b);
    }
}

// entries[(j << (w-1)) + e - 1] = e * bases[j] (projective), e in 1..2^(w-1)
template <class C>
__global__ void __launch_bounds__(BLOCK) k_table_entries(const uint32_t* bases, uint32_t* entries, int w, int nwin) {
    using G = Group<C>;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t half = (size_t)1 << (w - 1);
    if (tid >= half * nwin) return;
    int j = (int)(tid >> (w - 1));
    uint32_t e = (uint32_t)(tid & (half - 1)) + 1;
    Fe<C::N> b = G::curve_b();
    Proj<C> base = load_proj<C>(bases, j);
    Proj<C> acc = base;
    int top = 31 - __clz(e);
    for (int bit = top - 1; bit >= 0; bit--) {
        acc = G::dbl(acc, b);
        if ((e >> bit) & 1) acc = G::add(acc, base, b);
    }
    store_proj<C>(entries, tid, acc);
}

// ---- normalisation: (X:Y:Z) -> (X/Z, Y/Z) with Montgomery's trick --------------------------------
// `BatchNormalize::batch_normalize` (k256 projective.rs:367-391 + field.rs:244-265; primeorder
// projective.rs:452-478).  Thread t owns points t, t+T, t+2T, ... so that a wave always touches
// consecutive records; one field inversion per thread amortised over its K = n/T points.
// OUT_INTERNAL = false: big-endian canonical x||y records + identity flags (wire format)
// OUT_INTERNAL = true : [n][2][N] internal-form limbs (table entries; identities not expected)
template <class C, bool OUT_INTERNAL>
__global__ void __launch_bounds__(BLOCK) k_normalize(const uint32_t* proj, uint32_t* prefix, size_t n, size_t nthreads,
                            uint8_t* out_xy, uint8_t* out_inf, uint32_t* out_limbs) {
    using F = Field<C>;
    constexpr int N = C::N;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    Fe<N> acc = F::one();
    for (size_t j = t; j < n; j += nthreads) {
        Fe<N> z;
        load_limbs_vec<N>(z.v, proj + j * (3 * N) + 2 * N);
        store_limbs_vec<N>(prefix + j * N, acc.v);
        if (!F::is_zero(z)) acc = F::mul(acc, z);
    }
    Fe<N> inv = F::inv(acc);
    // walk back: last owned index first
    if (n <= t) return;
    size_t last = t + ((n - 1 - t) / nthreads) * nthreads;
    for (size_t j = last;; j -= nthreads) {
        Proj<C> p = load_proj<C>(proj, j);
        if (F::is_zero(p.z)) {
            if constexpr (!OUT_INTERNAL) {
                uint32_t zero[2 * N];
#pragma unroll
                for (int i = 0; i < 2 * N; i++) zero[i] = 0;
                store_limbs_vec<2 * N>(reinterpret_cast<uint32_t*>(out_xy + j * (8 * N)), zero);
                if (out_inf) out_inf[j] = 1;
            }
        } else {
            Fe<N> pre;
            load_limbs_vec<N>(pre.v, prefix + j * N);
            Fe<N> zinv = F::mul(pre, inv);
            inv = F::mul(inv, p.z);
            Fe<N> x = F::mul(p.x, zinv), y = F::mul(p.y, zinv);
            if constexpr (OUT_INTERNAL) {
                store_limbs_vec<N>(out_limbs + j * (2 * N), x.v);
                store_limbs_vec<N>(out_limbs + j * (2 * N) + N, y.v);
            } else {
                Fe<N> cx = F::to_canonical(x), cy = F::to_canonical(y);
                store_be_vec<N>(out_xy + j * (8 * N), cx.v);
                store_be_vec<N>(out_xy + j * (8 * N) + 4 * N, cy.v);
                if (out_inf) out_inf[j] = 0;
            }
        }
        if (j < nthreads) break;
    }
}

// ---- fixed base: out[i] = k[i] * G -----------------------------------------------------------------
// Drop-in for `mul_by_generator` (k256 mul.rs:180-197; primeorder basepoint.rs:82-99).  The reference
// walks 65 signed nibbles over a 33x8 projective table with full additions; here each lane walks
// nwin = bits/W + 1 signed W-bit windows over the affine table with complete *mixed* additions
// (RCB Alg 8 / Alg 5), so there are no doublings and no exceptional cases at all.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_fixed_base(const uint8_t* __restrict__ scalars, size_t n, const uint32_t* __restrict__ table, int w, int nwin,
             uint32_t* __restrict__ proj_out, int* status) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[N];
    load_scalar<C>(k, scalars, i, status);
    Fe<N> b = G::curve_b();
    Proj<C> acc = G::identity();
    uint32_t carry = 0;
    const size_t half = (size_t)1 << (w - 1);
#pragma unroll 1
    for (int j = 0; j < nwin; j++) {
        int d = signed_window_step(get_bits<N>(k, j * w, w), w, &carry);
        if (d != 0) {
            uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            const uint32_t* e = table + ((size_t)j * half + (mag - 1)) * (2 * N);
            Affine<C> q;
            load_limbs_vec<N>(q.x.v, e);
            load_limbs_vec<N>(q.y.v, e + N);
            if (d < 0) q.y = F::neg(q.y);
            acc = G::add_mixed(acc, q, b);
        }
    }
    store_proj<C>(proj_out, i, acc);
}

// ---- helpers for batch_normalize / point_sum ----------------------------------------------------------

// wire-format projective records (X||Y||Z big-endian canonical) -> internal projective scratch
template <class C>
__global__ void __launch_bounds__(BLOCK) k_load_proj(const uint8_t* xyz, size_t n, uint32_t* proj_out, int* status) {
    using F = Field<C>;
    constexpr int N = C::N;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Proj<C> p;
    Fe<N> c;
    bool ok = true;
    load_be_vec<N>(c.v, xyz + i * (12 * N));           ok = ok && !mp_geq<N>(c.v, C::P); p.x = F::from_canonical(c);
    load_be_vec<N>(c.v, xyz + i * (12 * N) + 4 * N);   ok = ok && !mp_geq<N>(c.v, C::P); p.y = F::from_canonical(c);
    load_be_vec<N>(c.v, xyz + i * (12 * N) + 8 * N);   ok = ok && !mp_geq<N>(c.v, C::P); p.z = F::from_canonical(c);
    if (!ok) atomicOr(status, ST_BAD_POINT);
    store_proj<C>(proj_out, i, p);
}

// workgroup-wide sum of one projective point per lane (LDS tree); result valid in lane 0
template <class C>
__device__ __forceinline__ Proj<C> block_sum(Proj<C> acc, uint32_t* lds, const Fe<C::N>& b) {
    using G = Group<C>;
    constexpr int N = C::N;
    uint32_t* mine = lds + threadIdx.x * (3 * N);
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
#pragma unroll
        for (int l = 0; l < N; l++) { mine[l] = acc.x.v[l]; mine[N + l] = acc.y.v[l]; mine[2 * N + l] = acc.z.v[l]; }
        __syncthreads();
        if ((int)threadIdx.x < s) {
            const uint32_t* o = lds + (threadIdx.x + s) * (3 * N);
            Proj<C> q;
#pragma unroll
            for (int l = 0; l < N; l++) { q.x.v[l] = o[l]; q.y.v[l] = o[N + l]; q.z.v[l] = o[2 * N + l]; }
            acc = G::add(acc, q, b);
        }
        __syncthreads();
    }
    return acc;
}

// out = sum of n affine points: one workgroup; lanes take strided subsets, then an LDS tree.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_point_sum(const uint8_t* points_xy, const uint8_t* points_inf, size_t n, uint32_t* proj_out, int* status) {
    using G = Group<C>;
    constexpr int N = C::N;
    __shared__ uint32_t lds[BLOCK * 3 * N];
    Fe<N> b = G::curve_b();
    Proj<C> acc = G::identity();
    for (size_t i = threadIdx.x; i < n; i += BLOCK) {
        Affine<C> a;
        if (load_affine<C>(&a, points_xy, points_inf, i, b, status)) acc = G::add_mixed(acc, a, b);
    }
    acc = block_sum<C>(acc, lds, b);
    if (threadIdx.x == 0) store_proj<C>(proj_out, 0, acc);
}

// pa[i] = pa[i] + pb[i]
template <class C>
__global__ void __launch_bounds__(BLOCK) k_proj_add_pairs(uint32_t* pa, const uint32_t* pb, size_t n) {
    using G = Group<C>;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<C::N> b = G::curve_b();
    store_proj<C>(pa, i, G::add(load_proj<C>(pa, i), load_proj<C>(pb, i), b));
}

}  // namespace ecgpu
