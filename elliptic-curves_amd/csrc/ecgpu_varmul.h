// ecgpu_varmul.h — one variable-base scalar multiplication k*P, the per-lane body of k_var_base (host +
// device; tests/hostcheck runs exactly this code on the CPU).
//
// Drop-in for `ProjectivePoint * Scalar` (primeorder/src/projective.rs:133-137 -> lincomb :532-557,
// LookupTable primeorder/src/tables/lookup.rs:30-38): table [P..8P], signed radix-16 digits d_i in [-8, 7]
// (Radix16Msb, bit-identical to Radix16Decomposition), most significant first, 4 doublings + 1 table
// addition per digit.
//
// Where the reference spends complete projective formulas on all of it, this ladder runs in Jacobian
// coordinates with incomplete formulas and switches to the complete addition for the last digit only.
// Why that is exact for every scalar 0 <= k < n and every finite P (all three groups have prime order n):
//   * let A_j = sum_{i >= j} d_i 16^(i-j) be the prefix value after digit j.  |sum_{i<j} d_i 16^i| < 16^j, so
//     A_j >= 0, A_j = 0 iff all digits processed so far are zero (the `started` flag), and
//     A_j <= k / 16^j + 1.
//   * doublings: the accumulator is A_j * P with 0 < A_j * 2^s < n, never the identity.
//   * addition at digit j >= 1: acc = m*P with m = 16 * A_(j+1), 16 <= m <= n/16 + 16; the table operand
//     is +-d*P with 1 <= d <= 8.  m = +-d (mod n) would need m = d or m = n - d; both are out of range.
//     So acc != +-operand and neither is the identity: the incomplete addition is exact.
//   * digit 0: m can reach n - d (e.g. k = n - 2 on a curve with n = 1 mod 16 gives acc = -P, operand -P).
//     This one addition converts both operands to homogeneous coordinates and uses the complete formula.
//   * table: e*P = (e-1)*P + P for e = 3..8 never has (e-1)*P = +-P; 2P is a doubling.
#pragma once

#include "ecgpu_point.h"
#include "ecgpu_recode.h"

namespace ecgpu {

// TabIO: void put(int e, const JacTab<C>&);  JacTab<C> get(int e) const;   e = 0..7 holds (e+1)*P
template <class C, class TabIO>
ECGPU_HD void var_build_table(const Affine<C>& a, TabIO& tab) {
    using G = Group<C>;
    Jac<C> t = G::jac_from_affine(a);
    const JacTab<C> t1 = G::jac_tab(t);
    tab.put(0, t1);
#pragma unroll 1
    for (int e = 1; e < 8; e++) {
        if (e == 1) t = G::jac_dbl(t);
        else t = G::jac_add(t, t1, false);
        tab.put(e, G::jac_tab(t));
    }
}

template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul_plain(const Affine<C>& a, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    using G = Group<C>;
    constexpr int N = C::N;
    var_build_table<C>(a, tab);
    Radix16Msb<N> digits;
    digits.init(k);
    Jac<C> acc = G::jac_from_affine(a);   // placeholder until the first non-zero digit
    bool started = false;
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
            JacTab<C> q = tab.get((d < 0 ? -d : d) - 1);
            if (started) {
                acc = G::jac_add(acc, q, d < 0);
            } else {
                acc = G::jac_from_tab(q, d < 0);
                started = true;
            }
        }
    }
    Proj<C> r = started ? G::jac_to_proj(acc) : G::identity();
    if (d != 0) {
        JacTab<C> q = tab.get((d < 0 ? -d : d) - 1);
        r = G::add(r, G::jac_tab_to_proj(q), b, d < 0);
    }
    return r;
}

// ---- k256: the same ladder on the GLV halves ---------------------------------------------------------------
// k256/src/arithmetic/mul.rs:112-163 (`lincomb` with one term), mul/glv.rs:149-156: k = r1 + r2*lambda (mod n) with
// |r1|, |r2| < 2^128 after folding the signs into the points, lambda*(x, y) = (beta*x, y).  Two 33-digit
// recodings share ONE chain of 128 doublings; the table of lambda*P is the table of P with X multiplied by beta on
// the fly (in Jacobian coordinates too: (X : Y : Z) -> (beta X : Y : Z), Z^2 and Z^3 unchanged).
//
// Exactness of the incomplete additions: after the doublings of digit j the accumulator is m*P with
// m = a + b*lambda, a = 16*A1, b = 16*A2 the prefixes of the two recodings (|a|, |b| <= 2^128 / 16^j + 16), and the
// operand is d*P or d*lambda*P, |d| <= 8.  acc = +-operand would make (a -+ d, b) resp. (a, b -+ d) a non-zero
// vector of the lattice {(x, y) : x + y*lambda = 0 mod n}, whose shortest vector has length 2^127.8 — impossible
// while both coordinates are below 2^125, i.e. for every digit j >= 1.  (The same argument gives acc != identity
// once a non-zero digit has been seen.)  Digit 0 uses the complete formula for both additions.
template <class TabIO>
ECGPU_HD Proj<K256Params> var_base_mul_glv(const Affine<K256Params>& a, const uint32_t* k,
                                           const Fe<K256Params::NL>& b, TabIO& tab) {
    using C = K256Params;
    using G = Group<C>;
    using F = Field<C>;
    var_build_table<C>(a, tab);
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
            JacTab<C> q = tab.get((e < 0 ? -e : e) - 1);
            if (half) q.x = F::mul(G::mt(q.x), beta).e;
            const bool neg = (e < 0) != (half ? s2 : s1);
            if (started) {
                acc = G::jac_add(acc, q, neg);
            } else {
                acc = G::jac_from_tab(q, neg);
                started = true;
            }
        }
    }
    Proj<C> r = started ? G::jac_to_proj(acc) : G::identity();
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        const int e = half ? e2 : e1;
        if (e == 0) continue;
        JacTab<C> q = tab.get((e < 0 ? -e : e) - 1);
        if (half) q.x = F::mul(G::mt(q.x), beta).e;
        r = G::add(r, G::jac_tab_to_proj(q), b, (e < 0) != (half ? s2 : s1));
    }
    return r;
}

template <class C, class TabIO>
ECGPU_HD Proj<C> var_base_mul(const Affine<C>& a, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    if constexpr (C::ID == CURVE_K256) return var_base_mul_glv(a, k, b, tab);
    else return var_base_mul_plain<C>(a, k, b, tab);
}

}  // namespace ecgpu
