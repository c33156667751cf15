// ecgpu_kernels.h — gfx950 kernels of the batch scalar-mul engine (HIP only).
//
// Data layout in HBM
//   scalars / points in, affine points out : the wire format of include/ecgpu.h (big-endian records of L
//       resp. 2L bytes; little-endian for bign256).  One lane owns one record; a wave touches 64 consecutive records = one contiguous
//       2 KiB / 4 KiB span, loaded and stored as 16-byte vectors and byte-swapped in registers.
//   "raw" field element      NS = ceil(NL/4)*4 u32 (12 for k256 9x29 and p256 10x28, 16 for p384 15x27): the
//       in-register limbs as they are (lazy form), padded so that records stay 16-byte vectors
//   "packed" field element   N u32 (8 / 8 / 12): the internal-domain value fully reduced to [0, p)
//   projective scratch       [n][3] raw elements
//   basepoint table          [nwin][2^(W-1)][2] packed elements, affine: entry (j, e) = e * 2^(W j) * G
//                            (64 B per entry for the 256-bit curves: 21.5 GB at k256's default W = 26)
//   variable-base table      [wave][8][3 NL][64] u32 (ecgpu_var.h): multiples 1..8 of each lane's point in AFFINE form
//                            (x, y; the third element holds the Z ratio while the table is built); a wave's access
//                            to one limb is 256 contiguous bytes
#pragma once

#include <hip/hip_runtime.h>

#include "ecgpu_point.h"
#include "ecgpu_recode.h"
#include "ecgpu_fixedmul.h"

namespace ecgpu {

enum : int { ST_BAD_SCALAR = 1, ST_BAD_POINT = 2 };

constexpr int BLOCK = 256;

// ---- 16-byte vector access helpers ---------------------------------------------------------------

// (NW a multiple of 4: 16-byte accesses; otherwise — the 28-byte records and 56-byte packed points of p224 — 32-bit ones)
template <int NW>
__device__ __forceinline__ void load_words_vec(uint32_t* dst, const uint32_t* src) {
    if constexpr (NW % 4 == 0) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
        for (int i = 0; i < NW / 4; i++) {
            uint4 v = s[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) dst[i] = src[i];
    }
}
// The same as exactly NW / 4 aligned 16-byte loads, for data that is used at once (the table-entry gather of k_fixed_base): every
// loaded quad passes through an empty asm statement as four live registers, so the compiler cannot re-cut the record.  Left to
// itself it turned the 64-byte entry into seven overlapping loads at 8-byte offsets once the k256 reduction's register window
// (ecgpu_k256_reduce_asm.h) changed the allocation around it.  (The wait for the data sits at the asm statement: not for loads
// that are meant to stay in flight, like the accumulation loop's prefetch.)
template <int NW>
__device__ __forceinline__ void load_words_vec_exact(uint32_t* dst, const uint32_t* src) {
    static_assert(NW % 4 == 0, "whole 16-byte pieces");
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4 v[NW / 4];
#pragma unroll
    for (int i = 0; i < NW / 4; i++) v[i] = s[i];
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
        dst[4 * i] = v[i].x; dst[4 * i + 1] = v[i].y; dst[4 * i + 2] = v[i].z; dst[4 * i + 3] = v[i].w;
    }
}
template <int NW>
__device__ __forceinline__ void store_words_vec(uint32_t* dst, const uint32_t* src) {
    if constexpr (NW % 4 == 0) {
        uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int i = 0; i < NW / 4; i++) d[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) dst[i] = src[i];
    }
}
// big-endian record of NW words -> little-endian words
template <int NW>
__device__ __forceinline__ void load_be_vec(uint32_t* words, const uint8_t* bytes) {
    uint32_t w[NW];
    load_words_vec<NW>(w, reinterpret_cast<const uint32_t*>(bytes));
#pragma unroll
    for (int i = 0; i < NW; i++) words[i] = bswap32(w[NW - 1 - i]);
}
template <int NW>
__device__ __forceinline__ void store_be_vec(uint8_t* bytes, const uint32_t* words) {
    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) w[NW - 1 - i] = bswap32(words[i]);
    store_words_vec<NW>(reinterpret_cast<uint32_t*>(bytes), w);
}

// one wire record (big-endian field element or scalar of WireBytes<C> bytes) <-> N little-endian words.  Records of
// 4 N bytes go through the vector helpers; others (p521: 66 bytes, 2-byte aligned) byte by byte.
template <class C>
__device__ __forceinline__ void load_wire(uint32_t* words, const uint8_t* bytes) {
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WireLe<C>::value) {                  // little-endian records (bignp256): the words as they lie
        load_words_vec<N>(words, reinterpret_cast<const uint32_t*>(bytes));
    } else if constexpr (WB == 4 * N) {
        load_be_vec<N>(words, bytes);
    } else {
        // 4 (N - 1) + 2 bytes (p521: 66).  Records are 2-byte aligned (WB is even, the bases are 16-byte aligned), so the
        // record is ONE halfword — the top word — followed by N - 1 big-endian words, read as explicit 16-bit pieces.
        // The first version read byte by byte; the compiler merges such loads into wide ones and extracts the bytes with
        // perm / SDWA sequences, and for some kernel shapes (k_selftest_field<P521Params>, then
        // k_ecdsa_recover_prepare<P521Params>: profiles/r02/diag_recover_p521.txt) the extracted operand came out with wrong
        // bits on gfx950 while the host build of the same source was right.  Whole-halfword pieces leave nothing to extract.
        static_assert(WB == 4 * (N - 1) + 2, "wire records are whole words or whole words + 2 bytes");
        const uint16_t* h = reinterpret_cast<const uint16_t*>(bytes);
        const uint32_t top = h[0];
        words[N - 1] = ((top & 0xffu) << 8) | (top >> 8);
#pragma unroll
        for (int m = 0; m < N - 1; m++) words[N - 2 - m] = bswap32((uint32_t)h[2 * m + 1] | ((uint32_t)h[2 * m + 2] << 16));
    }
}
template <class C>
__device__ __forceinline__ void store_wire(uint8_t* bytes, const uint32_t* words) {
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WireLe<C>::value) {
        store_words_vec<N>(reinterpret_cast<uint32_t*>(bytes), words);
    } else if constexpr (WB == 4 * N) {
        store_be_vec<N>(bytes, words);
    } else {
        static_assert(WB == 4 * (N - 1) + 2, "wire records are whole words or whole words + 2 bytes");
        uint16_t* h = reinterpret_cast<uint16_t*>(bytes);               // see load_wire
        const uint32_t top = words[N - 1];
        h[0] = (uint16_t)(((top & 0xffu) << 8) | ((top >> 8) & 0xffu));
#pragma unroll
        for (int m = 0; m < N - 1; m++) {
            const uint32_t d = bswap32(words[N - 2 - m]);
            h[2 * m + 1] = (uint16_t)d;
            h[2 * m + 2] = (uint16_t)(d >> 16);
        }
    }
}
template <class C>
__device__ __forceinline__ void zero_wire(uint8_t* bytes, int records) {
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WB == 4 * N) {
        uint32_t zero[N];
#pragma unroll
        for (int i = 0; i < N; i++) zero[i] = 0;
        for (int r = 0; r < records; r++) store_words_vec<N>(reinterpret_cast<uint32_t*>(bytes + r * WB), zero);
    } else {
        for (int j = 0; j < records * WB; j++) bytes[j] = 0;
    }
}
template <class C>
__device__ __forceinline__ void copy_wire(uint8_t* dst, const uint8_t* src) {   // bytes stay in wire order
    constexpr int WB = WireBytes<C>::value, N = C::N;
    if constexpr (WB == 4 * N) {
        uint32_t x[N];
        load_words_vec<N>(x, reinterpret_cast<const uint32_t*>(src));
        store_words_vec<N>(reinterpret_cast<uint32_t*>(dst), x);
    } else {
#pragma unroll
        for (int j = 0; j < WB; j++) dst[j] = src[j];
    }
}

// raw element <-> registers
template <class C>
__device__ __forceinline__ void store_raw(uint32_t* dst, const Fe<C::NL>& e) {
    constexpr int NS = Field<C>::NS;
    uint32_t w[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) w[i] = i < C::NL ? e.v[i] : 0u;
    store_words_vec<NS>(dst, w);
}
template <class C>
__device__ __forceinline__ Fe<C::NL> load_raw(const uint32_t* src) {
    constexpr int NS = Field<C>::NS;
    uint32_t w[NS];
    load_words_vec<NS>(w, src);
    Fe<C::NL> e;
#pragma unroll
    for (int i = 0; i < C::NL; i++) e.v[i] = w[i];
    return e;
}
template <class C>
__device__ __forceinline__ void store_proj(uint32_t* base, size_t idx, const Proj<C>& p) {
    constexpr int NS = Field<C>::NS;
    uint32_t* d = base + idx * (3 * NS);
    store_raw<C>(d, p.x);
    store_raw<C>(d + NS, p.y);
    store_raw<C>(d + 2 * NS, p.z);
}
template <class C>
__device__ __forceinline__ Proj<C> load_proj(const uint32_t* base, size_t idx) {
    constexpr int NS = Field<C>::NS;
    Proj<C> p;
    const uint32_t* s = base + idx * (3 * NS);
    p.x = load_raw<C>(s);
    p.y = load_raw<C>(s + NS);
    p.z = load_raw<C>(s + 2 * NS);
    return p;
}
// The same records QUAD-MAJOR ("limb-major" hand-over between k_fixed_base and k_normalize): 16-byte piece q of record idx at
// base + (q * n + idx) * 4 words, so that the 64 lanes of a wave, which own 64 consecutive records, read or write 1,024 contiguous
// bytes per load / store instruction — the record-major form above puts the lanes 3 NS words apart (144 bytes for k256: one
// instruction touches ~72 cache lines).  NQ pieces per record: 3 NS / 4 for a projective point, NS / 4 for one raw element.
template <int NQ>
__device__ __forceinline__ void store_quads_soa(uint32_t* base, size_t n, size_t idx, int q0, const uint32_t* w) {
#pragma unroll
    for (int q = 0; q < NQ; q++)
        *reinterpret_cast<uint4*>(base + ((size_t)(q0 + q) * n + idx) * 4) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
template <int NQ>
__device__ __forceinline__ void load_quads_soa(uint32_t* w, const uint32_t* base, size_t n, size_t idx, int q0) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + ((size_t)(q0 + q) * n + idx) * 4);
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
}
// element `comp` (0 = X, 1 = Y, 2 = Z) of record idx
template <class C>
__device__ __forceinline__ void store_raw_soa(uint32_t* base, size_t n, size_t idx, int comp, const Fe<C::NL>& e) {
    constexpr int NS = Field<C>::NS;
    uint32_t w[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) w[i] = i < C::NL ? e.v[i] : 0u;
    store_quads_soa<NS / 4>(base, n, idx, comp * (NS / 4), w);
}
template <class C>
__device__ __forceinline__ Fe<C::NL> load_raw_soa(const uint32_t* base, size_t n, size_t idx, int comp) {
    constexpr int NS = Field<C>::NS;
    uint32_t w[NS];
    load_quads_soa<NS / 4>(w, base, n, idx, comp * (NS / 4));
    Fe<C::NL> e;
#pragma unroll
    for (int i = 0; i < C::NL; i++) e.v[i] = w[i];
    return e;
}
template <class C>
__device__ __forceinline__ void store_proj_soa(uint32_t* base, size_t n, size_t idx, const Proj<C>& p) {
    store_raw_soa<C>(base, n, idx, 0, p.x);
    store_raw_soa<C>(base, n, idx, 1, p.y);
    store_raw_soa<C>(base, n, idx, 2, p.z);
}
template <class C>
__device__ __forceinline__ Proj<C> load_proj_soa(const uint32_t* base, size_t n, size_t idx) {
    Proj<C> p;
    p.x = load_raw_soa<C>(base, n, idx, 0);
    p.y = load_raw_soa<C>(base, n, idx, 1);
    p.z = load_raw_soa<C>(base, n, idx, 2);
    return p;
}
// packed affine point (2 x N words) <-> registers
template <class C>
__device__ __forceinline__ Affine<C> load_packed_affine(const uint32_t* src) {
    using F = Field<C>;
    uint32_t w[C::N];
    Affine<C> a;
    load_words_vec<C::N>(w, src);
    a.x = F::unpack(w).e;
    load_words_vec<C::N>(w, src + C::N);
    a.y = F::unpack(w).e;
    return a;
}
template <class C>
__device__ __forceinline__ void store_packed_affine(uint32_t* dst, const Fe<C::NL>& x, const Fe<C::NL>& y) {
    using F = Field<C>;
    uint32_t w[C::N];
    F::pack(w, Group<C>::m(x));
    store_words_vec<C::N>(dst, w);
    F::pack(w, Group<C>::m(y));
    store_words_vec<C::N>(dst + C::N, w);
}

// scalar record -> 32-bit words, flags out-of-range scalars (Scalar::from_repr, k256 scalar.rs:310-316)
template <class C>
__device__ __forceinline__ void load_scalar(uint32_t* k, const uint8_t* scalars, size_t i, int* status) {
    load_wire<C>(k, scalars + i * WireBytes<C>::value);
    if (mp_geq<C::N>(k, C::ORDER)) atomicOr(status, ST_BAD_SCALAR);
}

// affine point record -> internal form; returns false for the identity.  Flags coordinates >= p
// and off-curve points (AffinePoint::from_coordinates, primeorder/src/affine.rs:100-109).
// (cx, cy: the canonical coordinate words as read — for the plain-residue field of k256 they ARE the packed storage form)
template <class C>
__device__ __forceinline__ bool load_affine_words(Affine<C>* a, uint32_t* cx, uint32_t* cy, const uint8_t* xy, const uint8_t* inf,
                                                  size_t i, const Fe<C::NL>& b, int* status) {
    using F = Field<C>;
    if (inf != nullptr && inf[i]) return false;
    load_wire<C>(cx, xy + i * (2 * WireBytes<C>::value));
    load_wire<C>(cy, xy + i * (2 * WireBytes<C>::value) + WireBytes<C>::value);
    bool ok = !mp_geq<C::N>(cx, C::P) && !mp_geq<C::N>(cy, C::P);
    a->x = F::from_canonical(cx).e;
    a->y = F::from_canonical(cy).e;
    ok = ok && Group<C>::on_curve(*a, b);
    if (!ok) atomicOr(status, ST_BAD_POINT);
    return true;
}
template <class C>
__device__ __forceinline__ bool load_affine(Affine<C>* a, const uint8_t* xy, const uint8_t* inf, size_t i,
                                            const Fe<C::NL>& b, int* status) {
    uint32_t cx[C::N], cy[C::N];
    return load_affine_words<C>(a, cx, cy, xy, inf, i, b, status);
}

// ---- basepoint table construction -------------------------------------------------------------------
// Replaces the lazily built `BasepointTable` (primeorder/src/tables/basepoint.rs:41-76, k256
// tables.rs:11-18: 33/49 LUTs of 8 projective multiples) by one signed-window comb table of affine
// entries sized for HBM / Infinity Cache instead of a CPU L1.

// bases[j] = 2^(W*j) * G, j < nwin (one lane; nwin*W doublings)
template <class C>
__global__ void __launch_bounds__(64) k_window_bases(uint32_t* bases, int w, int nwin) {
    using G = Group<C>;
    using F = Field<C>;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    // wave-uniform data would send the whole chain to the scalar ALU (see k_msm_combine): an opaque VGPR zero added to
    // the generator keeps it on the vector ALU
    uint32_t vzero = 0;
    asm volatile("" : "+v"(vzero));
    uint32_t gx[C::N], gy[C::N];
#pragma unroll
    for (int i = 0; i < C::N; i++) { gx[i] = C::GX[i] + vzero; gy[i] = C::GY[i] + vzero; }
    Affine<C> g;
    g.x = F::from_canonical(gx).e;
    g.y = F::from_canonical(gy).e;
    Fe<C::NL> b = G::curve_b();
    Proj<C> p = G::from_affine(g);
    for (int j = 0; j < nwin; j++) {
        store_proj<C>(bases, j, p);
        for (int s = 0; s < w; s++) p = G::dbl(p, b);
    }
}

// entries[(j << (w-1)) + e - 1] = e * bases[j] (projective), e in 1..2^(w-1).
// Lane t of window j owns e = t + 1, t + 1 + T, t + 1 + 2T, ... (T = 2^tlog lanes per window, so that a wave always
// stores consecutive entries): the first one by double-and-add, the others by adding T * bases[j] (tlog doublings of
// the base) — about 3.5k instructions per entry at 64 entries per lane instead of 60k for double-and-add everywhere.
template <class C>
__global__ void __launch_bounds__(BLOCK) k_table_entries(const uint32_t* bases, uint32_t* entries, int w, int nwin, int tlog) {
    using G = Group<C>;
    const size_t T = (size_t)1 << tlog, half = (size_t)1 << (w - 1);
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= T * nwin) return;
    const int j = (int)(tid >> tlog);
    const uint32_t t = (uint32_t)(tid & (T - 1));
    Fe<C::NL> b = G::curve_b();
    Proj<C> base = load_proj<C>(bases, j);
    Proj<C> step = base;
    for (int s = 0; s < tlog; s++) step = G::dbl(step, b);
    const uint32_t e0 = t + 1;
    Proj<C> acc = base;
    for (int bit = 30 - __clz(e0); bit >= 0; bit--) {
        acc = G::dbl(acc, b);
        if ((e0 >> bit) & 1) acc = G::add(acc, base, b);
    }
    for (size_t e = e0; e <= half; e += T) {
        store_proj<C>(entries, (size_t)j * half + e - 1, acc);
        acc = G::add(acc, step, b);
    }
}

// ---- normalisation: (X:Y:Z) -> (X/Z, Y/Z) with Montgomery's trick --------------------------------
// `BatchNormalize::batch_normalize` (k256 projective.rs:367-391 + field.rs:244-265; primeorder
// projective.rs:452-478).  Lane t owns points t, t+T, t+2T, ... so that a wave always touches
// consecutive records; one field inversion per lane amortised over its K = n/T points.
// MODE NORM_WIRE      : big-endian canonical x||y records + identity flags (wire format)
// MODE NORM_PACKED    : [n][2] packed elements (table entries; identities not expected)
// MODE NORM_COMPRESSED: SEC1 compressed form split in two arrays: x (L bytes, to out_xy) and the tag byte 0x02 / 0x03
//                       (y even / odd), 0x00 for the identity (to out_inf) — `ToSec1Point::to_sec1_point(true)`,
//                       primeorder/src/affine.rs:387-401
enum : int { NORM_WIRE = 0, NORM_PACKED = 1, NORM_COMPRESSED = 2 };
// SOA: `proj` and `prefix` are quad-major (store_proj_soa; what k_fixed_base<C, true> leaves) instead of record-major.
template <class C, int MODE, bool SOA = false>
__global__ void __launch_bounds__(BLOCK) k_normalize(const uint32_t* proj, uint32_t* prefix, size_t n, size_t nthreads,
                                                     uint8_t* out_xy, uint8_t* out_inf, uint32_t* out_packed) {
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N, NS = F::NS, WB = WireBytes<C>::value;
    (void)N;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    // Both passes are software-pipelined by hand: the loads of the lane's NEXT point are issued before the multiplications of
    // the current one.  With one wave per SIMD (the launch is sized that way: one inversion per lane) nothing else hides a
    // load's ~2 us, and the loop-carried product keeps the compiler from hoisting the loads itself.
    const auto load_z = [&](size_t j) -> Fe<C::NL> {
        if constexpr (SOA) return load_raw_soa<C>(proj, n, j, 2);
        else return load_raw<C>(proj + j * (3 * NS) + 2 * NS);
    };
    const auto load_point = [&](size_t j) -> Proj<C> {
        if constexpr (SOA) return load_proj_soa<C>(proj, n, j);
        else return load_proj<C>(proj, j);
    };
    const auto load_prefix = [&](uint32_t* w, size_t j) {
        if constexpr (SOA) load_quads_soa<NS / 4>(w, prefix, n, j, 0);
        else load_words_vec<NS>(w, prefix + j * NS);
    };
    typename F::M1 acc = F::one();
    Fe<C::NL> z_next = load_z(t < n ? t : 0);
    for (size_t j = t; j < n; j += nthreads) {
        const Fe<C::NL> z = z_next;
        if (j + nthreads < n) z_next = load_z(j + nthreads);
        const bool ident = F::is_zero(G::m(z));
        {   // running product before this point, and whether the point is the identity, in the spare word
            static_assert(NS > C::NL, "raw form has no spare word");
            uint32_t w[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) w[i] = i < C::NL ? acc.e.v[i] : 0u;
            w[NS - 1] = ident ? 1u : 0u;
            if constexpr (SOA) store_quads_soa<NS / 4>(prefix, n, j, 0, w);
            else store_words_vec<NS>(prefix + j * NS, w);
        }
        if (!ident) acc = F::mul(acc, G::m(z));
    }
    typename F::M1 inv = F::inv(acc);
    if (n <= t) return;
    size_t last = t + ((n - 1 - t) / nthreads) * nthreads;
    Proj<C> p_next = load_point(last);
    uint32_t pw_next[NS];
    load_prefix(pw_next, last);
    for (size_t j = last;; j -= nthreads) {
        const Proj<C> p = p_next;
        uint32_t pw[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) pw[i] = pw_next[i];
        if (j >= nthreads) {
            p_next = load_point(j - nthreads);
            load_prefix(pw_next, j - nthreads);
        }
        if (pw[NS - 1]) {
            if constexpr (MODE == NORM_WIRE) {
                zero_wire<C>(out_xy + j * (2 * WB), 2);
                if (out_inf) out_inf[j] = 1;
            } else if constexpr (MODE == NORM_COMPRESSED) {
                zero_wire<C>(out_xy + j * WB, 1);
                out_inf[j] = 0;
            }
        } else {
            Fe<C::NL> pre_e;
#pragma unroll
            for (int i = 0; i < C::NL; i++) pre_e.v[i] = pw[i];
            typename F::M1 pre = G::m(pre_e);
            typename F::M1 zinv = F::mul(pre, inv);
            inv = F::mul(inv, G::m(p.z));
            typename F::M1 x = F::mul(G::m(p.x), zinv), y = F::mul(G::m(p.y), zinv);
            if constexpr (MODE == NORM_PACKED) {
                store_packed_affine<C>(out_packed + j * (2 * N), x.e, y.e);
            } else if constexpr (MODE == NORM_COMPRESSED) {
                uint32_t w[N];
                F::to_canonical(w, x);
                store_wire<C>(out_xy + j * WB, w);
                F::to_canonical(w, y);
                out_inf[j] = (uint8_t)(2u + (w[0] & 1u));
            } else {
                uint32_t w[N];
                F::to_canonical(w, x);
                store_wire<C>(out_xy + j * (2 * WB), w);
                F::to_canonical(w, y);
                store_wire<C>(out_xy + j * (2 * WB) + WB, w);
                if (out_inf) out_inf[j] = 0;
            }
        }
        if (j < nthreads) break;
    }
}

// ---- fixed base: out[i] = k[i] * G -----------------------------------------------------------------
// One lane per scalar; the algorithm and the reference citations are in ecgpu_fixedmul.h.
template <class C>
struct BaseTableHbm {
    const uint32_t* table;    // [nwin][2^(w-1)][2] packed elements
    size_t half;
    __device__ void load(PackedPoint<2 * C::N>& p, int window, uint32_t index) const {
        if constexpr ((2 * C::N) % 4 == 0) load_words_vec_exact<2 * C::N>(p.w, table + ((size_t)window * half + index) * (2 * C::N));
        else load_words_vec<2 * C::N>(p.w, table + ((size_t)window * half + index) * (2 * C::N));
    }
};
// (k256: three workgroups per CU = 156 registers; compiled for four — 128 registers — the kernel spills 340 bytes per lane)
template <class C, bool SOA = false>
__global__ void __launch_bounds__(BLOCK, C::A_IS_ZERO ? 3 : 1)
k_fixed_base(const uint8_t* __restrict__ scalars, size_t n, const uint32_t* __restrict__ table, int w, int nwin,
             uint32_t* __restrict__ proj_out, int* status) {
    using G = Group<C>;
    constexpr int N = C::N;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[N];
    load_scalar<C>(k, scalars, i, status);
    BaseTableHbm<C> tab{table, (size_t)1 << (w - 1)};
    const Proj<C> r = fixed_base_mul<C>(k, tab, w, nwin, G::curve_b());
    if constexpr (SOA) store_proj_soa<C>(proj_out, n, i, r);      // quad-major: what k_normalize<C, NORM_WIRE, true> reads
    else store_proj<C>(proj_out, i, r);
}

// ---- helpers for batch_normalize / point_sum ----------------------------------------------------------

// wire-format projective records (X||Y||Z big-endian canonical) -> internal projective scratch
template <class C>
__global__ void __launch_bounds__(BLOCK) k_load_proj(const uint8_t* xyz, size_t n, uint32_t* proj_out, int* status) {
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Proj<C> p;
    uint32_t c[N];
    bool ok = true;
    load_wire<C>(c, xyz + i * (3 * WB));           ok = ok && !mp_geq<N>(c, C::P); p.x = F::from_canonical(c).e;
    load_wire<C>(c, xyz + i * (3 * WB) + WB);   ok = ok && !mp_geq<N>(c, C::P); p.y = F::from_canonical(c).e;
    load_wire<C>(c, xyz + i * (3 * WB) + 2 * WB);   ok = ok && !mp_geq<N>(c, C::P); p.z = F::from_canonical(c).e;
    if (!ok) atomicOr(status, ST_BAD_POINT);
    store_proj<C>(proj_out, i, p);
}

// workgroup-wide sum of one projective point per lane (LDS tree); result valid in lane 0
template <class C>
__device__ __forceinline__ Proj<C> block_sum(Proj<C> acc, uint32_t* lds, const Fe<C::NL>& b) {
    using G = Group<C>;
    constexpr int NL = C::NL;
    uint32_t* mine = lds + threadIdx.x * (3 * NL);
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
#pragma unroll
        for (int l = 0; l < NL; l++) { mine[l] = acc.x.v[l]; mine[NL + l] = acc.y.v[l]; mine[2 * NL + l] = acc.z.v[l]; }
        __syncthreads();
        if ((int)threadIdx.x < s) {
            const uint32_t* o = lds + (threadIdx.x + s) * (3 * NL);
            Proj<C> q;
#pragma unroll
            for (int l = 0; l < NL; l++) { q.x.v[l] = o[l]; q.y.v[l] = o[NL + l]; q.z.v[l] = o[2 * NL + l]; }
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
    __shared__ uint32_t lds[BLOCK * 3 * C::NL];
    Fe<C::NL> b = G::curve_b();
    Proj<C> acc = G::identity();
    for (size_t i = threadIdx.x; i < n; i += BLOCK) {
        Affine<C> a;
        if (load_affine<C>(&a, points_xy, points_inf, i, b, status)) acc = G::add_mixed(acc, a, b);
    }
    acc = block_sum<C>(acc, lds, b);
    if (threadIdx.x == 0) store_proj<C>(proj_out, 0, acc);
}

// out[g] = sum of in[g * BLOCK .. g * BLOCK + BLOCK) (the identity past n): one level of a reduction tree over
// projective points, one workgroup per output
template <class C>
__global__ void __launch_bounds__(BLOCK) k_proj_sum_level(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ out) {
    using G = Group<C>;
    __shared__ uint32_t lds[BLOCK * 3 * C::NL];
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    Proj<C> acc = i < n ? load_proj<C>(in, i) : G::identity();
    acc = block_sum<C>(acc, lds, G::curve_b());
    if (threadIdx.x == 0) store_proj<C>(out, blockIdx.x, acc);
}

}  // namespace ecgpu
