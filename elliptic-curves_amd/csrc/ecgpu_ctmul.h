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
//                                       (LUTs of 2^(8 i) G, even nibbles into acc, odd nibbles into acc2, acc + 16 acc2)
//
// What "uniform schedule" means here, and what tools/ct_isa_check.py verifies on the gfx950 ISA of the two kernels:
//   * the number of digits is fixed (8 N + 1 radix-16 digits of `Radix16Decomposition`, 33 per GLV half), zero digits are
//     not skipped, the accumulator starts at the identity, every digit step is a COMPLETE addition and the doublings between have no exceptional
//     case (complete ones for k256, Jacobian ones with the identity patched under a mask elsewhere: ct_dbl4);
//   * a table entry is picked by reading ALL EIGHT entries and keeping one under a mask (`v_bfi_b32` under an opaque mask), the sign of a digit
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
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        Proj<C> e;
        e.x = tab.get_el(j, 0);
        e.y = tab.get_el(j, 1);
        e.z = tab.get_el(j, 2);
#pragma unroll
        for (int h = 0; h < H; h++) t[h] = ct_sel_proj<C>(xabs[h] == (uint32_t)(j + 1), e, t[h]);
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
        if (di != 8 * N) acc = ct_dbl4<C>(acc, b);
        bool neg;
        const uint32_t xabs = ct_abs_digit(digits.digit(di), &neg);
        Proj<C> t;
        ct_table_scan<C, TabIO, 1>(tab, &xabs, &t);
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
        if (di != 32) acc = ct_dbl4<C>(acc, b);
        bool neg[2];
        uint32_t xabs[2];
        xabs[0] = ct_abs_digit(d1.digit(di), &neg[0]);
        xabs[1] = ct_abs_digit(d2.digit(di), &neg[1]);
        Proj<C> t[2];
        ct_table_scan<C, TabIO, 2>(tab, xabs, t);
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
// LUT i (i < CT_BASE_LUTS) holds e * 2^(8 i) * G, e = 1..8, affine, packed: `BasepointTable::new`
// (primeorder/src/tables/basepoint.rs:41-76; k256/src/arithmetic/tables.rs:11-18) with affine instead of projective
// entries (the additions are mixed complete ones, 11M instead of 12M, and an entry is 2 instead of 3 elements to scan).
// A zero digit has no affine entry: the addition is carried out against entry 1 and its result dropped under a mask.
template <class C>
constexpr int CT_BASE_LUTS = 4 * C::N + 1;          // 33 for 256-bit scalars, 49 for 384-bit ones (basepoint.rs: 1 + bytes)

// Lut: void load(PackedPoint<2N>&, int lut, int entry) const — entry (entry + 1) * 2^(8 lut) * G
template <class C, class Lut>
ECGPU_HD Proj<C> ct_lut_add(const Proj<C>& acc, const Lut& lut, int i, int digit, const Fe<C::NL>& b) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    bool neg;
    const uint32_t xabs = ct_abs_digit(digit, &neg);
    PackedPoint<2 * N> sel;
    lut.load(sel, i, 0);                            // entry 1: also what a zero digit adds (and drops)
#pragma unroll 1
    for (int j = 1; j < 8; j++) {
        PackedPoint<2 * N> e;
        lut.load(e, i, j);
        const uint32_t hit = ct_mask(xabs == (uint32_t)(j + 1));
#pragma unroll
        for (int w = 0; w < 2 * N; w++) sel.w[w] = ct_pick(hit, e.w[w], sel.w[w]);
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
    constexpr int N = C::N, NLUT = CT_BASE_LUTS<C>;
    Radix16Msb<N> digits;
    digits.init(k);
    Proj<C> acc = ct_lut_add<C>(G::identity(), lut, NLUT - 1, digits.digit(8 * N), b);
    Proj<C> acc2 = G::identity();
#pragma unroll 1
    for (int i = NLUT - 2; i >= 0; i--) {
        acc2 = ct_lut_add<C>(acc2, lut, i, digits.digit(2 * i + 1), b);
        acc = ct_lut_add<C>(acc, lut, i, digits.digit(2 * i), b);
    }
#pragma unroll 1
    for (int s = 0; s < 4; s++) acc2 = G::dbl(acc2, b);
    return G::add(acc, acc2, b);
}

}  // namespace ecgpu
