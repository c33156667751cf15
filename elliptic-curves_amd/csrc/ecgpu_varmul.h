// ecgpu_varmul.h — one variable-base scalar multiplication k*P, the per-lane body of k_var_base (host +
// device; tests/hostcheck runs exactly this code on the CPU).
//
// Drop-in for `ProjectivePoint * Scalar` (primeorder/src/projective.rs:133-137 -> lincomb :532-557,
// LookupTable primeorder/src/tables/lookup.rs:30-38): table [P..8P], signed radix-16 digits d_i in [-8, 7]
// (Radix16Msb, bit-identical to Radix16Decomposition), most significant first, 4 doublings + 1 table
// addition per digit.
//
// Where the reference spends complete projective formulas on all of it, this ladder runs in Jacobian
// coordinates with incomplete formulas over an affine table and switches to the complete (mixed) addition for the
// last digit only.
// Why that is exact for every scalar 0 <= k < n and every finite P (all three groups have prime order n):
//   * let A_j = sum_{i >= j} d_i 16^(i-j) be the prefix value after digit j.  |sum_{i<j} d_i 16^i| < 16^j, so
//     A_j >= 0, A_j = 0 iff all digits processed so far are zero (the `started` flag), and
//     A_j <= k / 16^j + 1.
//   * doublings: the accumulator is A_j * P with 0 < A_j * 2^s < n, never the identity.
//   * addition at digit j >= 1: acc = m*P with m = 16 * A_(j+1), 16 <= m <= n/16 + 16; the table operand
//     is +-d*P with 1 <= d <= 8.  m = +-d (mod n) would need m = d or m = n - d; both are out of range.
//     So acc != +-operand and neither is the identity: the incomplete addition is exact.
//   * digit 0: m can reach n - d (e.g. k = n - 2 on a curve with n = 1 mod 16 gives acc = -P, operand -P).
//     This one addition converts the accumulator to homogeneous coordinates and uses the complete formula.
//   * table: e*P = (e-1)*P + P for e = 3..8 never has (e-1)*P = +-P; 2P is a doubling.
#pragma once

#include "ecgpu_point.h"
#include "ecgpu_recode.h"

// -DECGPU_VAR_PREFETCH=0: the k256 GLV ladder as round 4 had it (table entries fetched in front of their additions) — A/B:
// profiles/r05/var_ladder_table_prefetch_ab.txt
#ifndef ECGPU_VAR_PREFETCH
#define ECGPU_VAR_PREFETCH 1
#endif

namespace ecgpu {

// Table chain shared by both ladders: entry e (0..7) = (e + 1) P.  2P is a doubling, (e + 1) P = e P + P a mixed addition
// whose result has Z_(e+1) = Z_e * H_e, so the ratio between the Z of consecutive entries comes for free.  Stores
// (X_e, Y_e, Z_e / Z_(e-1)) in elements 0..2 of entry e and returns Z_8.
// TabIO: put_el(entry, k, element) / get_el(entry, k), entries 0..7, k = 0..2.
template <class C, class TabIO>
ECGPU_HD Fe<C::NL> var_table_chain(const Affine<C>& a, TabIO& tab) {
    using G = Group<C>;
    Jac<C> t = G::jac_from_affine(a);
    tab.put_el(0, 0, a.x);
    tab.put_el(0, 1, a.y);
    t = G::jac_dbl(t);                          // Z_2 / Z_1 = Z_2
    tab.put_el(1, 0, t.x);
    tab.put_el(1, 1, t.y);
    tab.put_el(1, 2, t.z);
#pragma unroll 1
    for (int e = 2; e < 8; e++) {
        Fe<C::NL> h;
        t = G::jac_madd(t, a, false, &h);
        tab.put_el(e, 0, t.x);
        tab.put_el(e, 1, t.y);
        tab.put_el(e, 2, h);
    }
    return t.z;
}

// Affine table: one field inversion per lane (Z_8, by division steps: about seven additions' worth of work) and a
// backward pass over the Z ratios (r = r * ratio, x = X r^2, y = Y r^3) turn the chain into [P..8P] in affine
// coordinates.  Every ladder addition is then a mixed one (8M + 3S instead of 11M + 3S with cached Z^2, Z^3), an entry
// is 2 elements instead of 5 to store and to fetch back 64 (96) times, and the last, complete addition is a mixed one too.
template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul_plain(const Affine<C>& a, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    {
        const Fe<C::NL> zg = var_table_chain<C>(a, tab);
        typename F::M1 r = F::inv(G::m(zg));          // Z_8 is a product: magnitude 1
#pragma unroll 1
        for (int e = 7; e >= 1; e--) {          // entry 0 is P itself
            if (e < 7) r = F::mul(r, G::mj(tab.get_el(e + 1, 2)));
            auto r2 = F::sqr(r);
            tab.put_el(e, 0, F::mul(G::mj(tab.get_el(e, 0)), r2).e);
            tab.put_el(e, 1, F::mul(G::mj(tab.get_el(e, 1)), F::mul(r2, r)).e);
        }
    }
    Radix16Msb<N> digits;
    digits.init(k);
    Jac<C> acc = G::jac_from_affine(a);   // placeholder until the first non-zero digit
    bool started = false;
    // (Requesting a digit's table entry before the four doublings that precede its addition was measured for these ladders in round 5
    // and changes nothing — p256 15.75 against 15.71 ms, p384 43.95 against 43.85, same box, alternating: with two waves per SIMD the
    // 2 NL strided loads are not what a wave waits for.  The GLV ladder below, with two entries per step, gains 1.9 % and keeps it.)
    int d = 0;
#pragma unroll 1
    for (int di = 8 * N; di >= 0; di--) {
        if (started) {
#pragma unroll 1
            for (int s = 0; s < 4; s++) acc = G::jac_dbl(acc);
        }
        d = digits.digit(di);
        if (di == 0) break;
        if (d != 0) {
            const int idx = (d < 0 ? -d : d) - 1;
            Affine<C> q;
            q.x = tab.get_el(idx, 0);
            q.y = tab.get_el(idx, 1);
            if (started) {
                acc = G::jac_madd(acc, q, d < 0);
            } else {
                if (d < 0) q.y = G::neg_coord(q.y);
                acc = G::jac_from_affine(q);
                started = true;
            }
        }
    }
    Proj<C> r = started ? G::jac_to_proj(acc) : G::identity();
    if (d != 0) {
        const int idx = (d < 0 ? -d : d) - 1;
        Affine<C> q;
        q.x = tab.get_el(idx, 0);
        q.y = tab.get_el(idx, 1);
        r = G::add_mixed(r, q, b, d < 0);
    }
    return r;
}

// ---- k256: the same ladder on the GLV halves, over a shared-Z table ------------------------------------------------
// k256/src/arithmetic/mul.rs:112-163 (`lincomb` with one term), mul/glv.rs:149-156: k = r1 + r2*lambda (mod n) with
// |r1|, |r2| < 2^128 after folding the signs into the points, lambda*(x, y) = (beta*x, y).  Two 33-digit
// recodings share ONE chain of 128 doublings.
//
// Shared-Z table.  e*P = (e-1)*P + P is a mixed addition whose result has Z_e = Z_(e-1) * H_e, so the ratios between
// the Z of consecutive entries come for free; one backward pass (r = r * H, x' = X r^2, y' = Y r^3: 5M per entry)
// rewrites every entry over the Z of the last one, Zg = Z_8.  (x'_e, y'_e) is then e*P's image under the isomorphism
// (x, y) -> (x Zg^2, y Zg^3) onto y^2 = x^3 + 7 Zg^6, an a = 0 curve again: the doublings do not see the difference, the
// table is AFFINE there, and every ladder addition is a mixed one (8M + 3S instead of 11M + 3S with cached Z^2, Z^3).
// beta*x' is still the x of lambda*(e*P).  Mapping back multiplies the accumulator's Z by Zg.  The table costs what the
// Jacobian one did (six mixed instead of full additions pay for the backward pass) and holds 2 instead of 5 elements
// per entry.
//
// Exactness of the incomplete additions: after the doublings of digit j the accumulator is m*P with
// m = a + b*lambda, a = 16*A1, b = 16*A2 the prefixes of the two recodings (|a|, |b| <= 2^128 / 16^j + 16), and the
// operand is d*P or d*lambda*P, |d| <= 8.  acc = +-operand would make (a -+ d, b) resp. (a, b -+ d) a non-zero
// vector of the lattice {(x, y) : x + y*lambda = 0 mod n}, whose shortest vector has length 2^127.8 — impossible
// while both coordinates are below 2^125, i.e. for every digit j >= 1.  (The same argument gives acc != identity
// once a non-zero digit has been seen.)  Digit 0 uses the complete formula for both additions, on the original curve.
template <class TabIO>
ECGPU_HD Proj<K256Params> var_base_mul_glv(const Affine<K256Params>& a, const uint32_t* k,
                                           const Fe<K256Params::NL>& b, TabIO& tab) {
    using C = K256Params;
    using G = Group<C>;
    using F = Field<C>;
    using E = Fe<C::NL>;
    {
        const E zg = var_table_chain<C>(a, tab);
        typename F::M1 r = F::one();
#pragma unroll 1
        for (int e = 6; e >= 0; e--) {
            r = F::mul(r, G::mj(tab.get_el(e + 1, 2)));
            auto r2 = F::sqr(r);
            tab.put_el(e, 0, F::mul(G::mj(tab.get_el(e, 0)), r2).e);
            tab.put_el(e, 1, F::mul(G::mj(tab.get_el(e, 1)), F::mul(r2, r)).e);
        }
        tab.put_el(7, 2, zg);                   // kept for the way back
    }
    uint32_t r1[8], r2[8];
    K256Scalar::decompose(r1, r2, k);
    const bool s1 = K256Scalar::is_high(r1), s2 = K256Scalar::is_high(r2);
    if (s1) K256Scalar::neg(r1, r1);
    if (s2) K256Scalar::neg(r2, r2);
    Radix16Msb<5> d1, d2;                       // |r_i| < 2^129: 5 words, digits 0..33
    d1.init(r1);
    d2.init(r2);
    const typename F::M1 beta = F::unpack(C::BETA);
    Jac<C> acc = G::jac_from_affine(a);         // placeholder until the first non-zero digit
    bool started = false;
    int e1 = 0, e2 = 0;
#if ECGPU_VAR_PREFETCH
    // Both halves' table entries are requested BEFORE the four doublings that precede their additions (round 5): unconditionally — a
    // zero digit asks for entry 0 and does not use it —, so that the 4 NL strided loads of two per-lane entries have four doublings to
    // arrive in: 8.42 -> 8.26 ms per 2^20 multiplications, although 56 more registers go to scratch around the table build
    Affine<C> q1, q2;
    auto fetch = [&](Affine<C>& q, int e) {
        const int idx = e == 0 ? 0 : (e < 0 ? -e : e) - 1;
        q.x = tab.get_el(idx, 0);
        q.y = tab.get_el(idx, 1);
    };
    e1 = d1.digit(33);
    e2 = d2.digit(33);
    fetch(q1, e1);
    fetch(q2, e2);
#pragma unroll 1
    for (int di = 33;; di--) {
        if (started) {
#pragma unroll 1
            for (int s = 0; s < 4; s++) acc = G::jac_dbl(acc);
        }
        if (di == 0) break;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {  // one copy of the addition code for both halves
            const int e = half ? e2 : e1;
            if (e == 0) continue;
            Affine<C> q = half ? q2 : q1;
            if (half) q.x = F::mul(G::mj(q.x), beta).e;
            const bool neg = (e < 0) != (half ? s2 : s1);
            if (started) {
                acc = G::jac_madd(acc, q, neg);
            } else {
                if (neg) q.y = G::neg_coord(q.y);
                acc = G::jac_from_affine(q);
                started = true;
            }
        }
        e1 = d1.digit(di - 1);
        e2 = d2.digit(di - 1);
        fetch(q1, e1);
        fetch(q2, e2);
    }
#else
#pragma unroll 1
    for (int di = 33; di >= 0; di--) {
        if (started) {
#pragma unroll 1
            for (int s = 0; s < 4; s++) acc = G::jac_dbl(acc);
        }
        e1 = d1.digit(di);
        e2 = d2.digit(di);
        if (di == 0) break;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {  // one copy of the addition code for both halves
            const int e = half ? e2 : e1;
            if (e == 0) continue;
            const int idx = (e < 0 ? -e : e) - 1;
            Affine<C> q;
            q.x = tab.get_el(idx, 0);
            q.y = tab.get_el(idx, 1);
            if (half) q.x = F::mul(G::mj(q.x), beta).e;
            const bool neg = (e < 0) != (half ? s2 : s1);
            if (started) {
                acc = G::jac_madd(acc, q, neg);
            } else {
                if (neg) q.y = G::neg_coord(q.y);
                acc = G::jac_from_affine(q);
                started = true;
            }
        }
    }
#endif
    // back to the original curve: Z * Zg; the table entries are (x' : y' : Zg) there
    const auto zg = G::mj(tab.get_el(7, 2));
    Proj<C> r = G::identity();
    if (started) {
        acc.z = F::mul(G::mj(acc.z), zg).e;
        r = G::jac_to_proj(acc);
    }
    const auto zg3 = F::mul(F::sqr(zg), zg);
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int e = half ? e2 : e1;
        if (e == 0) continue;
        const int idx = (e < 0 ? -e : e) - 1;
        auto x = G::mj(tab.get_el(idx, 0));
        Proj<C> q;
        q.x = (half ? F::mul(F::mul(x, beta), zg) : F::mul(x, zg)).e;
        q.y = tab.get_el(idx, 1);
        q.z = zg3.e;
        r = G::add(r, q, b, (e < 0) != (half ? s2 : s1));
    }
    return r;
}

template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul(const Affine<C>& a, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    if constexpr (C::ID == CURVE_K256) return var_base_mul_glv(a, k, b, tab);
    else return var_base_mul_plain<C>(a, k, b, tab);
}

}  // namespace ecgpu
