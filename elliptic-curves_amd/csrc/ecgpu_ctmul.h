// ecgpu_ctmul.h — uniform-schedule ("constant-time shaped") scalar multiplications, the per-lane bodies of
// k_var_base_ct / k_fixed_base_ct (host + device; tests/hostcheck runs exactly this code on the CPU).
//
// These are the reference's constant-time drivers restated as they are, not the cheaper variable-time ladders of
// ecgpu_varmul.h / ecgpu_fixedmul.h:
//   * `ProjectivePoint * Scalar`        primeorder/src/projective.rs:133-137, 532-557 (`lincomb` with one term) over
//                                       `LookupTable::new` / `select` (primeorder/src/tables/lookup.rs:30-65)
//   * k256 `ProjectivePoint * Scalar`   k256/src/arithmetic/mul.rs:112-163 (`lincomb`: GLV halves, signs folded into the
//                                       tables, 33 digits each)
//   * `mul_by_generator`                k256/src/arithmetic/mul.rs:180-197, primeorder/src/tables/basepoint.rs:82-99
//                                       (LUTs of multiples of 2^(W i) G scanned in full — W = 6 here, 4 there: "generator" below)
//
// What "uniform schedule" means here, and what tools/ct_isa_check.py verifies on the gfx950 ISA of the two kernels:
//   * the number of digits is fixed (8 N + 1 radix-16 digits of `Radix16Decomposition`, 33 per GLV half; one 6-bit digit per
//     generator LUT), zero digits are
//     not skipped, the accumulator starts at the identity, every digit step is a COMPLETE addition and the doublings between have no exceptional
//     case (complete ones for k256, Jacobian ones with the identity patched under a mask elsewhere: ct_dbl4);
//   * a table entry is picked by reading ALL entries (8; 32 of a generator LUT) and keeping one under a mask (`v_bfi_b32` under an opaque mask), the sign of a digit
//     by a masked negation: no memory address and no branch condition is computed from scalar (or point) data — the only
//     conditional branches are the bounds checks on the lane index and the loop counters;
//   * range / on-curve verdicts are written as one flag byte per element and folded into the status word by a second
//     kernel (k_ct_flags): the check itself takes the same path for valid and invalid input.
// Not covered: k_normalize, which follows, branches on "result is the identity" (k = 0 or P = identity) and nothing else
// (its inversion is the branch-free division-step one, ecgpu_modinv.h).
#pragma once

#include "ecgpu_point.h"
#include "ecgpu_recode.h"

namespace ecgpu {

enum : uint32_t { CT_FLAG_BAD_SCALAR = 1, CT_FLAG_BAD_POINT = 2 };

// all ones / all zeros from a flag.  On the device the value is passed through an empty asm statement: the compiler may not
// know that it came from a comparison, so it cannot turn the masked selects below back into `flag ? a : b` and then a
// group of such selects under one flag into a conditional block of moves (it did: the first build of the k256 kernel
// had an exec-masked branch around the second select of the table scan — found by tools/ct_isa_check.py).
ECGPU_HD uint32_t ct_mask(bool flag) {
    uint32_t m = 0u - (uint32_t)flag;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    return m;
}
// a where the mask is set, b elsewhere (one v_bfi_b32)
ECGPU_HD uint32_t ct_pick(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }

// sel[w] = (x == want) ? e[w] : sel[w] for NW words — the inner statement of a table scan.  The portable form is ct_pick under
// one mask: three 32-bit logic instructions per word once the compiler is through with it (and, and, or: 1.5 issue slots).  On
// the device the same selection is written as v_cndmask_b32 in its 32-bit encoding, half a slot per word, in groups of up to
// eight words behind one v_cmp (an asm statement takes at most 30 operands).  All operands are vector registers: with vcc as
// the condition the instruction has no constant-bus slot left for a scalar source on gfx9-family targets (tried: the assembler
// refuses `v_cndmask_b32 v, s, v, vcc`) — the generator scan (ct_lut_scan_uniform) uses v_bfi_b32 for scalar sources.
// No branch, no address: nothing here for a compiler to turn into control flow, and nothing tools/ct_isa_check.py objects to.
#if defined(__HIP_DEVICE_COMPILE__)
template <int NW, int W0 = 0>
__device__ __forceinline__ void ct_pick_words(uint32_t* sel, const uint32_t* e, uint32_t x, uint32_t want) {
    if constexpr (NW - W0 >= 8) {
        asm("v_cmp_ne_u32 vcc, %8, %9\n\tv_cndmask_b32 %0, %10, %0, vcc\n\tv_cndmask_b32 %1, %11, %1, vcc\n\t"
                     "v_cndmask_b32 %2, %12, %2, vcc\n\tv_cndmask_b32 %3, %13, %3, vcc\n\tv_cndmask_b32 %4, %14, %4, vcc\n\t"
                     "v_cndmask_b32 %5, %15, %5, vcc\n\tv_cndmask_b32 %6, %16, %6, vcc\n\tv_cndmask_b32 %7, %17, %7, vcc"
                     : "+v"(sel[W0]), "+v"(sel[W0 + 1]), "+v"(sel[W0 + 2]), "+v"(sel[W0 + 3]), "+v"(sel[W0 + 4]), "+v"(sel[W0 + 5]),
                       "+v"(sel[W0 + 6]), "+v"(sel[W0 + 7])
                     : "s"(want), "v"(x), "v"(e[W0]), "v"(e[W0 + 1]), "v"(e[W0 + 2]), "v"(e[W0 + 3]), "v"(e[W0 + 4]), "v"(e[W0 + 5]),
                       "v"(e[W0 + 6]), "v"(e[W0 + 7])
                     : "vcc");
        ct_pick_words<NW, W0 + 8>(sel, e, x, want);
    } else if constexpr (NW - W0 >= 2) {
        asm("v_cmp_ne_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %5, %1, vcc"
                     : "+v"(sel[W0]), "+v"(sel[W0 + 1])
                     : "s"(want), "v"(x), "v"(e[W0]), "v"(e[W0 + 1])
                     : "vcc");
        ct_pick_words<NW, W0 + 2>(sel, e, x, want);
    } else if constexpr (NW - W0 == 1) {
        asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %0, vcc" : "+v"(sel[W0]) : "s"(want), "v"(x), "v"(e[W0]) : "vcc");
    }
}
#else
template <int NW, int W0 = 0>
inline void ct_pick_words(uint32_t* sel, const uint32_t* e, uint32_t x, uint32_t want) {
    const uint32_t hit = ct_mask(x == want);
    for (int w = W0; w < NW; w++) sel[w] = ct_pick(hit, e[w], sel[w]);
}
#endif

template <class C>
ECGPU_HD Fe<C::NL> ct_sel_fe(uint32_t m, const Fe<C::NL>& a, const Fe<C::NL>& b) {
    Fe<C::NL> r;
#pragma unroll
    for (int i = 0; i < C::NL; i++) r.v[i] = ct_pick(m, a.v[i], b.v[i]);
    return r;
}
template <class C>
ECGPU_HD Proj<C> ct_sel_proj(bool flag, const Proj<C>& a, const Proj<C>& b) {
    const uint32_t m = ct_mask(flag);
    Proj<C> r;
    r.x = ct_sel_fe<C>(m, a.x, b.x);
    r.y = ct_sel_fe<C>(m, a.y, b.y);
    r.z = ct_sel_fe<C>(m, a.z, b.z);
    return r;
}

// |d| and sign of a signed digit without a data-dependent branch (lookup.rs:47-49)
ECGPU_HD uint32_t ct_abs_digit(int d, bool* neg) {
    const int m = d >> 31;
    *neg = m != 0;
    return (uint32_t)((d + m) ^ m);
}

// TabIO: put_el(entry, k, element) / get_el(entry, k), entries 0..7, k = 0..2 (X, Y, Z of (entry + 1) P).
// `LookupTable::new`: points[j + 1] = p + points[j]  (lookup.rs:30-38), complete additions.
template <class C, class TabIO>
ECGPU_HD void ct_table_build(const Proj<C>& p, const Fe<C::NL>& b, TabIO& tab) {
    using G = Group<C>;
    Proj<C> t = p;
#pragma unroll 1
    for (int e = 0; e < 8; e++) {
        tab.put_el(e, 0, t.x);
        tab.put_el(e, 1, t.y);
        tab.put_el(e, 2, t.z);
        if (e < 7) t = G::add(t, p, b);
    }
}

// `LookupTable::select` for one or two digits in ONE pass over the eight entries: t_h = |d_h| P (the identity for
// d_h = 0); every entry is read whatever the digits are.
template <class C, class TabIO, int H>
ECGPU_HD void ct_table_scan(const TabIO& tab, const uint32_t* xabs, Proj<C>* t) {
    using G = Group<C>;
#pragma unroll
    for (int h = 0; h < H; h++) t[h] = G::identity();
#pragma unroll 2
    for (int j = 0; j < 8; j++) {
        Proj<C> e;
        e.x = tab.get_el(j, 0);
        e.y = tab.get_el(j, 1);
        e.z = tab.get_el(j, 2);
#pragma unroll
        for (int h = 0; h < H; h++) {
            ct_pick_words<C::NL>(t[h].x.v, e.x.v, xabs[h], (uint32_t)(j + 1));
            ct_pick_words<C::NL>(t[h].y.v, e.y.v, xabs[h], (uint32_t)(j + 1));
            ct_pick_words<C::NL>(t[h].z.v, e.z.v, xabs[h], (uint32_t)(j + 1));
        }
    }
}

// The four doublings between two digits.  a = 0 (k256): the complete doubling (6M + 2S) four times, as the reference does.
// The other curves: through Jacobian coordinates — (X : Y : Z) -> (X Z, Y Z^2, Z), four times dbl-2001-b / dbl-2007-bl
// (3M + 5S against 8M + 3S for the complete doubling), back with (X Z : Y : Z^3) — 7.3 M-equivalents instead of 10.5 per
// doubling, 6-7 for the two conversions.  The Jacobian doubling has no exceptional case on these curves (every finite point
// has odd order) and keeps (t^2, t^3, 0) at infinity; the one point the conversion cannot express, the identity (0 : Y : 0) ->
// (0, 0, 0), is replaced by (1, 1, 0) under a mask, so the schedule stays the same for every accumulator value.  The digit
// addition stays the complete one: it is where P + P, P - P and the identity occur.
template <class C>
ECGPU_HD Proj<C> ct_dbl4(const Proj<C>& acc, const Fe<C::NL>& b) {
    using G = Group<C>;
    using F = Field<C>;
    if constexpr (C::A_IS_ZERO) {
        Proj<C> r = acc;
#pragma unroll 1
        for (int s = 0; s < 4; s++) r = G::dbl(r, b);
        return r;
    } else {
        const auto X = G::m(acc.x), Y = G::m(acc.y), Z = G::m(acc.z);
        const uint32_t inf = ct_mask(F::is_zero(Z));
        const Fe<C::NL> one = F::one().e;
        typename G::J j;
        j.x = ct_sel_fe<C>(inf, one, F::mul(X, Z).e);
        j.y = ct_sel_fe<C>(inf, one, F::mul(Y, F::sqr(Z)).e);
        j.z = acc.z;
#pragma unroll 1
        for (int s = 0; s < 4; s++) j = G::jac_dbl(j);
        return G::jac_to_proj(j);
    }
}

// The four doublings AND the table scan for the digit that follows them, interleaved (round 5): the reads of entries 2s + 1, 2s + 2
// are issued in front of doubling s and selected behind it, so that they fly under ~850 multiply-adds instead of being waited
// for in a loop of their own between the doublings and the addition (the stand-alone scan's waits were what the second wave per
// SIMD had to cover: 84 % of the kernel's cycles issued an instruction against 89 % for the variable-time ladder).  The same
// entries, masks and selections in the same order for every input: nothing for tools/ct_isa_check.py to object to.  Costs the
// registers of two entries in flight across a doubling (2 x 3 NL) and of the selection (3 NL H): used where that fits (NL <= 10).
#ifndef ECGPU_CT_SCAN_UNDER_DBL
#define ECGPU_CT_SCAN_UNDER_DBL 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ECGPU_CT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ECGPU_CT_SCHED_FENCE() ((void)0)
#endif
template <class C>
constexpr bool ct_scan_under_dbl() { return ECGPU_CT_SCAN_UNDER_DBL && C::NL <= 10; }

template <class C, class TabIO, int H>
ECGPU_HD Proj<C> ct_dbl4_scan(const Proj<C>& acc, const Fe<C::NL>& b, const TabIO& tab, const uint32_t* xabs, Proj<C>* t) {
    using G = Group<C>;
    using F = Field<C>;
#pragma unroll
    for (int h = 0; h < H; h++) t[h] = G::identity();
    auto fetch = [&](int s, Proj<C>* e) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            e[u].x = tab.get_el(2 * s + u, 0);
            e[u].y = tab.get_el(2 * s + u, 1);
            e[u].z = tab.get_el(2 * s + u, 2);
        }
    };
    auto select = [&](int s, const Proj<C>* e) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int h = 0; h < H; h++) {
                ct_pick_words<C::NL>(t[h].x.v, e[u].x.v, xabs[h], (uint32_t)(2 * s + u + 1));
                ct_pick_words<C::NL>(t[h].y.v, e[u].y.v, xabs[h], (uint32_t)(2 * s + u + 1));
                ct_pick_words<C::NL>(t[h].z.v, e[u].z.v, xabs[h], (uint32_t)(2 * s + u + 1));
            }
    };
    if constexpr (C::A_IS_ZERO) {
        Proj<C> r = acc;
#pragma unroll 1
        for (int s = 0; s < 4; s++) {
            Proj<C> e[2];
            fetch(s, e);
            ECGPU_CT_SCHED_FENCE();
            r = G::dbl(r, b);
            ECGPU_CT_SCHED_FENCE();
            select(s, e);
        }
        return r;
    } else {
        const auto X = G::m(acc.x), Y = G::m(acc.y), Z = G::m(acc.z);
        const uint32_t inf = ct_mask(F::is_zero(Z));
        const Fe<C::NL> one = F::one().e;
        typename G::J j;
        j.x = ct_sel_fe<C>(inf, one, F::mul(X, Z).e);
        j.y = ct_sel_fe<C>(inf, one, F::mul(Y, F::sqr(Z)).e);
        j.z = acc.z;
#pragma unroll 1
        for (int s = 0; s < 4; s++) {
            Proj<C> e[2];
            fetch(s, e);
            ECGPU_CT_SCHED_FENCE();
            j = G::jac_dbl(j);
            ECGPU_CT_SCHED_FENCE();
            select(s, e);
        }
        return G::jac_to_proj(j);
    }
}

// p: the point in homogeneous coordinates ((0 : 1 : 0) for the identity), k: N words < n.
template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul_ct_plain(const Proj<C>& p, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    using G = Group<C>;
    constexpr int N = C::N;
    ct_table_build<C>(p, b, tab);
    Radix16Msb<N> digits;
    digits.init(k);
    Proj<C> acc = G::identity();
#pragma unroll 1
    for (int di = 8 * N; di >= 0; di--) {
        bool neg;
        const uint32_t xabs = ct_abs_digit(digits.digit(di), &neg);
        Proj<C> t;
        if constexpr (ct_scan_under_dbl<C>()) {
            if (di != 8 * N) acc = ct_dbl4_scan<C, TabIO, 1>(acc, b, tab, &xabs, &t);
            else ct_table_scan<C, TabIO, 1>(tab, &xabs, &t);
        } else {
            if (di != 8 * N) acc = ct_dbl4<C>(acc, b);
            ct_table_scan<C, TabIO, 1>(tab, &xabs, &t);
        }
        acc = G::add(acc, t, b, neg);
    }
    return acc;
}

// k256: k = r1 + r2 lambda, both halves folded to |r_i| < 2^128 with the signs moved into the table entries
// (mul.rs:112-137: `LookupTable::new(conditional_select(x, -x, r1_sign))`, the second table over the endomorphism image);
// here ONE table of P serves both halves: j (lambda P) = (beta X_j : Y_j : Z_j), and the fold sign joins the digit sign.
template <class TabIO>
ECGPU_HD Proj<K256Params> var_base_mul_ct_glv(const Proj<K256Params>& p, const uint32_t* k, const Fe<K256Params::NL>& b,
                                              TabIO& tab) {
    using C = K256Params;
    using G = Group<C>;
    using F = Field<C>;
    ct_table_build<C>(p, b, tab);
    uint32_t r1[8], r2[8], n1[8], n2[8];
    K256Scalar::decompose(r1, r2, k);
    const bool s1 = K256Scalar::is_high(r1), s2 = K256Scalar::is_high(r2);
    K256Scalar::neg(n1, r1);
    K256Scalar::neg(n2, r2);
    const uint32_t m1 = ct_mask(s1), m2 = ct_mask(s2);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r1[i] = ct_pick(m1, n1[i], r1[i]);
        r2[i] = ct_pick(m2, n2[i], r2[i]);
    }
    Radix16Msb<5> d1, d2;                       // |r_i| < 2^129: 5 words, digits 0..32 are the reference's 33
    d1.init(r1);
    d2.init(r2);
    const typename F::M1 beta = F::unpack(C::BETA);
    Proj<C> acc = G::identity();
#pragma unroll 1
    for (int di = 32; di >= 0; di--) {
        bool neg[2];
        uint32_t xabs[2];
        xabs[0] = ct_abs_digit(d1.digit(di), &neg[0]);
        xabs[1] = ct_abs_digit(d2.digit(di), &neg[1]);
        Proj<C> t[2];
        if constexpr (ct_scan_under_dbl<C>()) {
            if (di != 32) acc = ct_dbl4_scan<C, TabIO, 2>(acc, b, tab, xabs, t);
            else ct_table_scan<C, TabIO, 2>(tab, xabs, t);
        } else {
            if (di != 32) acc = ct_dbl4<C>(acc, b);
            ct_table_scan<C, TabIO, 2>(tab, xabs, t);
        }
        t[1].x = F::mul(G::m(t[1].x), beta).e;
        acc = G::add(acc, t[0], b, neg[0] != s1);
        acc = G::add(acc, t[1], b, neg[1] != s2);
    }
    return acc;
}

template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul_ct(const Proj<C>& p, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    if constexpr (C::ID == CURVE_K256) return var_base_mul_ct_glv(p, k, b, tab);
    else return var_base_mul_ct_plain<C>(p, k, b, tab);
}

// ---- generator ----------------------------------------------------------------------------------------------------
// The reference (`BasepointTable`, primeorder/src/tables/basepoint.rs:41-99; k256/src/arithmetic/mul.rs:180-197) keeps 33 LUTs of
// e 2^(8 i) G, e = 1..8, and adds one entry per nibble into two accumulators: 65 complete additions and four doublings.  The
// same idea with a wider window does less: one LUT PER WINDOW of W = 6 bits (LUT i = e 2^(6 i) G, e = 1..32, affine, packed),
// 43 instead of 65 additions for a 256-bit scalar and no doublings at all; scanning 32 entries instead of 8 costs 32 x 17
// select instructions per window against the ~1,800 of the addition it saves a third of (W = 5: 52 windows, W = 7: 37 windows
// of 64 entries — both cost more; first version, W = 4 as the reference: k256 4.6 ms per 2^20, profiles/r03/ct_rates_and_counters.txt).
// The additions are mixed complete ones (11M, entries of 2 instead of 3 elements).  A zero digit has no affine entry: the addition
// is carried out against entry 1 and its result dropped under a mask.
template <class C>
constexpr int ct_scalar_bits() {                     // bit length of the group order
    int b = 32;
    while (b > 1 && !((C::ORDER[C::N - 1] >> (b - 1)) & 1u)) b--;
    return 32 * (C::N - 1) + b;
}
template <class C>
constexpr int CT_BASE_LUTS = ct_scalar_bits<C>() / CT_BASE_W + 1;     // 43 for 256-bit orders, 65 for 384, 87 for 521

// Lut: void load(PackedPoint<2N>&, int lut, int entry) const — entry (entry + 1) * 2^(W lut) * G;  UNIFORM: the loads are at
// wave-uniform addresses, the words arrive in scalar registers
#if defined(__HIP_DEVICE_COMPILE__)
// The scan of a LUT that arrives through the scalar cache is bound by the latency of its loads, not by its instructions
// (profiles/r03/ct_generator_scan_variants.txt), and scalar loads return out of order: the only wait there is waits for ALL of
// them.  So the load of entry j + 1 is issued right AFTER the wait for entry j and flies under the selection of entry j: the
// first word of entry j is selected, then the next index is passed through an empty asm statement that names that word as an
// input (the load cannot be scheduled above the selection it seems to depend on), then the other words are selected.
template <class C, class Lut>
__device__ __forceinline__ void ct_lut_scan_uniform(PackedPoint<2 * C::N>& sel, const Lut& lut, int i, uint32_t xabs) {
    constexpr int NW = 2 * C::N;
    PackedPoint<NW> e[2];
    lut.load(sel, i, 0);
    lut.load(e[1], i, 1);
#pragma unroll
    for (int j = 1; j < CT_BASE_ENTRIES; j++) {
        const uint32_t hit = ct_mask(xabs == (uint32_t)(j + 1));
        const uint32_t* cur = e[j & 1].w;
        asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(sel.w[0]) : "v"(hit), "s"(cur[0]));
        if (j + 1 < CT_BASE_ENTRIES) {
            int jn = j + 1;
            asm volatile("" : "+s"(jn) : "v"(sel.w[0]));
            lut.load(e[(j + 1) & 1], i, jn);
            __builtin_amdgcn_sched_barrier(0);      // ... and not sunk below the selections that follow
        }
#pragma unroll
        for (int w = 1; w < NW; w++) asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(sel.w[w]) : "v"(hit), "s"(cur[w]));
    }
}
#endif

template <class C, class Lut>
ECGPU_HD Proj<C> ct_lut_add(const Proj<C>& acc, const Lut& lut, int i, int digit, const Fe<C::NL>& b) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    bool neg;
    const uint32_t xabs = ct_abs_digit(digit, &neg);
    PackedPoint<2 * N> sel;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (Lut::UNIFORM) {
        ct_lut_scan_uniform<C>(sel, lut, i, xabs);
    } else
#endif
    {
        lut.load(sel, i, 0);                        // entry 1: also what a zero digit adds (and drops)
#pragma unroll 2
        for (int j = 1; j < CT_BASE_ENTRIES; j++) {
            PackedPoint<2 * N> e;
            lut.load(e, i, j);
            ct_pick_words<2 * N>(sel.w, e.w, xabs, (uint32_t)(j + 1));
        }
    }
    Affine<C> q;
    q.x = F::unpack(sel.w).e;
    q.y = F::unpack(sel.w + N).e;
    const Proj<C> r = G::add_mixed(acc, q, b, neg);
    return ct_sel_proj<C>(xabs == 0, acc, r);
}

template <class C, class Lut>
ECGPU_HD Proj<C> fixed_base_mul_ct(const uint32_t* k, const Lut& lut, const Fe<C::NL>& b) {
    using G = Group<C>;
    using Digits = SignedWindowsMsb<C::N, CT_BASE_W, ct_scalar_bits<C>()>;
    static_assert(Digits::COUNT == CT_BASE_LUTS<C>, "one LUT per window");
    Digits digits;
    digits.init(k);
    Proj<C> acc = G::identity();
#pragma unroll 1
    for (int i = CT_BASE_LUTS<C> - 1; i >= 0; i--) acc = ct_lut_add<C>(acc, lut, i, digits.digit(i), b);
    return acc;
}

}  // namespace ecgpu
