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
ECGPU_HD Proj<C> var_base_mul(const Affine<C>& a, const uint32_t* k, const Fe<C::NL>& b, TabIO& tab) {
    using G = Group<C>;
    constexpr int N = C::N;
    Jac<C> t = G::jac_from_affine(a);
    const JacTab<C> t1 = G::jac_tab(t);
    tab.put(0, t1);
#pragma unroll 1
    for (int e = 1; e < 8; e++) {
        if (e == 1) t = G::jac_dbl(t);
        else t = G::jac_add(t, t1, false);
        tab.put(e, G::jac_tab(t));
    }
    Radix16Msb<N> digits;
    digits.init(k);
    Jac<C> acc = t;   // placeholder until the first non-zero digit
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

}  // namespace ecgpu
