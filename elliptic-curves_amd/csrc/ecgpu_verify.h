// ecgpu_verify.h — the per-element logic of the verification / decompression kernels on little-endian word arrays
// (host + device: the kernels of ecgpu_ecdsa.h wrap these in wire loads and stores, tests/hostcheck runs exactly this code
// on the CPU against the oracle).  The equations and the reference lines they follow are stated in ecgpu_ecdsa.h.
#pragma once

#include "ecgpu_point.h"
#include "ecgpu_scalar.h"

namespace ecgpu {

// x, y canonical candidates (any 32 N-bit values): a point of the curve with coordinates below p?
template <class C>
ECGPU_HD bool verify_point_ok(const uint32_t* cx, const uint32_t* cy) {
    using F = Field<C>;
    if (mp_geq<C::N>(cx, C::P) || mp_geq<C::N>(cy, C::P)) return false;
    Affine<C> a;
    a.x = F::from_canonical(cx).e;
    a.y = F::from_canonical(cy).e;
    return Group<C>::on_curve(a, Group<C>::curve_b());
}

// a failed element is fed to the arithmetic kernels as 0 * G + 0 * G
template <class C>
ECGPU_HD void verify_blank(bool ok, uint32_t* a, uint32_t* b, uint32_t* cx, uint32_t* cy) {
#pragma unroll
    for (int j = 0; j < C::N; j++) {
        a[j] = ok ? a[j] : 0u;
        b[j] = ok ? b[j] : 0u;
        cx[j] = ok ? cx[j] : C::GX[j];
        cy[j] = ok ? cy[j] : C::GY[j];
    }
}

// ECDSA (verify_prehashed): range checks, key validation, u1 = z / s, u2 = r / s.  Returns the element's validity.
template <class C>
// w_in: s^-1 mod n computed elsewhere (k_scalar_batch_inv: Montgomery's trick over the batch; 0 for an s outside [1, n - 1], which
// fails the range check here anyway), or nullptr: the inversion happens here.
ECGPU_HD bool ecdsa_prepare_words(const uint32_t* zw, const uint32_t* rw, const uint32_t* sw, uint32_t* cx, uint32_t* cy,
                                  int reject_high_s, uint32_t* u1, uint32_t* u2, const uint32_t* w_in = nullptr) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    bool ok = !S::is_zero(rw) && S::in_range(rw) && !S::is_zero(sw) && S::in_range(sw);
    if (reject_high_s) ok = ok && !S::is_high(sw);
    ok = verify_point_ok<C>(cx, cy) && ok;
    uint32_t zr[N], w[N];
    S::reduce_wire(zr, zw);
    if (w_in) {
#pragma unroll
        for (int j = 0; j < N; j++) w[j] = w_in[j];
    } else {
        S::inv(w, sw);
    }
    S::mul(u1, zr, w);
    S::mul(u2, rw, w);
    verify_blank<C>(ok, u1, u2, cx, cy);
    return ok;
}
// x(R) mod n == r   (x < p < 2 n)
template <class C>
ECGPU_HD bool ecdsa_finish_words(const uint32_t* x, const uint32_t* rw) {
    uint32_t xr[C::N];
    ScalarN<C>::reduce_once(xr, x);
    bool eq = true;
#pragma unroll
    for (int j = 0; j < C::N; j++) eq = eq && (xr[j] == rw[j]);
    return eq;
}

// SM2DSA on the prehash (sm2/src/dsa/verifying.rs:138-171): r, s in [1, n - 1], t = r + s mod n != 0, Q on the curve;
// a = s, b = t for (x1, y1) = s G + t Q
template <class C>
ECGPU_HD bool sm2dsa_prepare_words(const uint32_t* rw, uint32_t* sw, uint32_t* cx, uint32_t* cy, uint32_t* t) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    bool ok = !S::is_zero(rw) && S::in_range(rw) && !S::is_zero(sw) && S::in_range(sw);
    uint32_t sum[N], d[N];
    const uint32_t carry = mp_add<N>(sum, rw, sw);
    const uint32_t borrow = mp_sub<N>(d, sum, C::ORDER);
    const bool use_d = carry || !borrow;                      // r + s >= n (the inputs are below n when ok)
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = use_d ? d[j] : sum[j];
    ok = ok && !S::is_zero(t);
    ok = verify_point_ok<C>(cx, cy) && ok;
    verify_blank<C>(ok, sw, t, cx, cy);
    return ok;
}
// r == (e mod n) + (x1 mod n) mod n; x1 = 0 for the identity (`to_affine().x()` of the identity), as in the reference
template <class C>
ECGPU_HD bool sm2dsa_finish_words(const uint32_t* ew, const uint32_t* x, bool identity, const uint32_t* rw) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    uint32_t er[N], xr[N], sum[N], d[N];
    S::reduce_once(er, ew);
    S::reduce_once(xr, x);
#pragma unroll
    for (int j = 0; j < N; j++) xr[j] = identity ? 0u : xr[j];
    const uint32_t carry = mp_add<N>(sum, er, xr);
    const uint32_t borrow = mp_sub<N>(d, sum, C::ORDER);
    const bool use_d = carry || !borrow;
    bool eq = true;
#pragma unroll
    for (int j = 0; j < N; j++) eq = eq && ((use_d ? d[j] : sum[j]) == rw[j]);
    return eq;
}

// bign on the prehash (bignp256/src/ecdsa/verifying.rs:100-147 with `Signature::from_bytes`, bignp256/src/ecdsa.rs:72-88): the
// 48-byte signature is S0 (16 bytes) || S1 (32 bytes), both little-endian; S0 = 0, S1 = 0 or S1 >= q do not parse.
//     a = (S1 + H) mod q   (H = the 32 hash bytes as a little-endian integer, reduced: `Scalar::reduce`),   b = S0 + 2^128
// for R = a G + b Q.  (b needs no reduction: 2^129 < q.)
template <class C>
ECGPU_HD bool bign_prepare_words(const uint32_t* hw, const uint32_t* s0w, const uint32_t* s1w, uint32_t* cx, uint32_t* cy, uint32_t* a,
                                 uint32_t* b) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    static_assert(N == 8, "bign signatures are defined here for l = 128 (bign-curve256v1)");
    bool ok = (s0w[0] | s0w[1] | s0w[2] | s0w[3]) != 0u && !S::is_zero(s1w) && S::in_range(s1w);
    uint32_t hr[N], sum[N], d[N];
    S::reduce_once(hr, hw);                                  // H < 2^256 < 2 q
    const uint32_t carry = mp_add<N>(sum, s1w, hr);
    const uint32_t borrow = mp_sub<N>(d, sum, C::ORDER);
    const bool use_d = carry || !borrow;                     // S1 + H >= q (S1 is below q when ok)
#pragma unroll
    for (int j = 0; j < N; j++) {
        a[j] = use_d ? d[j] : sum[j];
        b[j] = j < 4 ? s0w[j] : (j == 4 ? 1u : 0u);
    }
    ok = verify_point_ok<C>(cx, cy) && ok;
    verify_blank<C>(ok, a, b, cx, cy);
    return ok;
}

// -e mod n of a challenge word array (e is reduced first)
template <class C>
ECGPU_HD void schnorr_neg_challenge(uint32_t* ne, const uint32_t* ew) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    uint32_t er[N], d[N];
    S::reduce_once(er, ew);
    bool z = S::is_zero(er);
    mp_sub<N>(d, C::ORDER, er);
#pragma unroll
    for (int j = 0; j < N; j++) ne[j] = z ? 0u : d[j];
}
// BIP340 with the challenge given: a = s, b = -e; r < p, 0 < s < n, P on the curve
template <class C>
ECGPU_HD bool schnorr_prepare_words(const uint32_t* ew, const uint32_t* rw, uint32_t* sw, uint32_t* cx, uint32_t* cy, uint32_t* ne) {
    using S = ScalarN<C>;
    bool ok = !mp_geq<C::N>(rw, C::P) && !S::is_zero(sw) && S::in_range(sw);
    ok = verify_point_ok<C>(cx, cy) && ok;
    schnorr_neg_challenge<C>(ne, ew);
    verify_blank<C>(ok, sw, ne, cx, cy);
    return ok;
}
// lift_x: cy = the even square root of cx^3 + 7 (k256); false for cx >= p or a non-residue
template <class C>
ECGPU_HD bool schnorr_lift_x(const uint32_t* cx, uint32_t* cy) {
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N;
    static_assert(C::A_IS_ZERO, "BIP340 is defined over secp256k1");
    bool ok = !mp_geq<N>(cx, C::P);
    auto x = F::from_canonical(cx);
    auto alpha = F::norm(F::add(F::mul(F::sqr(x), x), G::m(G::curve_b())));
    bool root;
    auto beta = F::sqrt(alpha, &root);
    ok = ok && root;
    F::to_canonical(cy, beta);
    if (cy[0] & 1u) {
        uint32_t d[N];
        mp_sub<N>(d, C::P, cy);
#pragma unroll
        for (int j = 0; j < N; j++) cy[j] = d[j];
    }
    return ok;
}
// R finite, y(R) even, x(R) == r
template <class C>
ECGPU_HD bool schnorr_finish_words(const uint32_t* x, const uint32_t* y, const uint32_t* rw) {
    bool eq = true;
#pragma unroll
    for (int j = 0; j < C::N; j++) eq = eq && (x[j] == rw[j]);
    return eq && !(y[0] & 1u);
}

// DecompressPoint::decompress: cy = the root of cx^3 + a cx + b with the requested parity; false (and zeroed cx, cy) for
// cx >= p or a non-residue
template <class C>
ECGPU_HD bool decompress_words(uint32_t* cx, bool y_is_odd, uint32_t* cy) {
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N;
    bool ok = !mp_geq<N>(cx, C::P);
    auto x = F::from_canonical(cx);
    auto x3 = F::mul(F::sqr(x), x);
    typename F::M1 alpha;
    if constexpr (C::A_IS_ZERO) {
        alpha = F::norm(F::add(x3, G::m(G::curve_b())));
    } else {
        if constexpr (GenericA<C>::value) {
            alpha = F::mul(F::norm(F::add(F::add(x3, F::mul(G::curve_a(), x)), G::m(G::curve_b()))), F::one());
        } else {
            auto x3x = F::add(F::dbl(x), x);
            alpha = F::mul(F::add(F::norm(F::sub(x3, x3x)), G::m(G::curve_b())), F::one());   // back to magnitude (1, 1)
        }
    }
    bool root;
    auto beta = F::sqrt(alpha, &root);
    ok = ok && root;
    F::to_canonical(cy, beta);
    if (((cy[0] & 1u) != 0) != y_is_odd) {                      // the other root: p - beta (beta != 0 here, or parity
        uint32_t d[N];                                          // 0 was asked for and beta = 0 stays)
        bool z = mp_is_zero<N>(cy);
        mp_sub<N>(d, C::P, cy);
#pragma unroll
        for (int j = 0; j < N; j++) cy[j] = z ? 0u : d[j];
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        cx[j] = ok ? cx[j] : 0u;
        cy[j] = ok ? cy[j] : 0u;
    }
    return ok;
}

// ECDSA public-key recovery, everything before the two scalar multiplications — `VerifyingKey::recover_from_prehash` of
// the `ecdsa` crate (0.17.0, un-vendored; the reference's vectors: k256/src/ecdsa.rs:190-262, p256/tests/ecdsa.rs:20-25):
// r, s in [1, n - 1] (and s <= (n - 1) / 2 where the final `verify_prehash` normalises s: k256), recovery id <= 3,
// x = r or r + n (bit 1; `checked_add`, and x >= p fails the decompression), R = decompress(x, bit 0),
// a = -(z / r), b = s / r for the key a G + b R.  What `verify_prehash` checks on the recovered key beyond the high-s rule
// holds by construction: (z / s) G + (r / s) (a G + b R) = R, whose x is r mod n, and R is a finite point.
template <class C>
ECGPU_HD bool ecdsa_recover_prepare_words(const uint32_t* zw, const uint32_t* rw, const uint32_t* sw, uint32_t recid,
                                          int reject_high_s, uint32_t* a, uint32_t* b, uint32_t* cx, uint32_t* cy,
                                          const uint32_t* rinv_in = nullptr) {      // (r^-1 from k_scalar_batch_inv, or nullptr)
    using S = ScalarN<C>;
    constexpr int N = C::N;
    bool ok = recid <= 3u && !S::is_zero(rw) && S::in_range(rw) && !S::is_zero(sw) && S::in_range(sw);
    if (reject_high_s) ok = ok && !S::is_high(sw);
    uint32_t sum[N];
    const uint32_t carry = mp_add<N>(sum, rw, C::ORDER);
    const bool reduced = (recid & 2u) != 0;
#pragma unroll
    for (int j = 0; j < N; j++) cx[j] = reduced ? sum[j] : rw[j];
    ok = ok && !(reduced && carry);
    ok = decompress_words<C>(cx, (recid & 1u) != 0, cy) && ok;
    uint32_t zr[N], rinv[N], t[N], d[N];
    S::reduce_wire(zr, zw);
    if (rinv_in) {
#pragma unroll
        for (int j = 0; j < N; j++) rinv[j] = rinv_in[j];
    } else {
        S::inv(rinv, rw);
    }
    S::mul(t, rinv, zr);
    const bool tz = S::is_zero(t);
    mp_sub<N>(d, C::ORDER, t);
#pragma unroll
    for (int j = 0; j < N; j++) a[j] = tz ? 0u : d[j];
    S::mul(b, rinv, sw);
    verify_blank<C>(ok, a, b, cx, cy);
    return ok;
}

}  // namespace ecgpu
