// ecgpu_ecdsa.h — batch ECDSA verification around the aG + bP kernels (HIP only).
//
// the dominant caller of `mul_by_generator_and_mul_add_vartime`
// (primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310).  The verification equation itself
// lives in the un-vendored `ecdsa` crate 0.17.0 (Cargo.lock:428-429, `hazmat::verify_prehashed`, instantiated at
// p256/src/ecdsa.rs:69,161-169, p384/src/ecdsa.rs:179-187, k256/src/ecdsa.rs:104-106); it is the published
// algorithm of SEC1 v2 §4.1.4 / FIPS 186-5 §6.4.2:
//     z = the leftmost bits of the digest as an integer, reduced mod n        (`Reduce<FieldBytes>`)
//     reject unless 1 <= r, s < n   (and, where the curve sets NORMALIZE_S, unless s <= (n-1)/2)
//     w = s^-1 mod n,  u1 = z w,  u2 = r w,  R = u1 G + u2 Q;  accept iff R != identity and x(R) mod n == r
// One lane per signature; an invalid element never fails the batch, it just gets ok = 0.
#pragma once

#include "ecgpu_kernels.h"
#include "ecgpu_scalar.h"
#include "ecgpu_sha256.h"

namespace ecgpu {

// prepare: range checks, public-key validation, u1 / u2 as wire-format scalars for the scalar-mul kernels.
// Elements that are already known to fail get u1 = u2 = 0 and Q = G so that the arithmetic kernels see valid input.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_prepare(const uint8_t* __restrict__ z, const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                const uint8_t* __restrict__ q_xy, size_t n, int reject_high_s, uint8_t* __restrict__ u1_out,
                uint8_t* __restrict__ u2_out, uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid) {
    using S = ScalarN<C>;
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t zw[N], rw[N], sw[N], cx[N], cy[N];
    load_wire<C>(zw, z + i * WB);
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    load_wire<C>(cx, q_xy + i * (2 * WB));
    load_wire<C>(cy, q_xy + i * (2 * WB) + WB);
    bool ok = !S::is_zero(rw) && S::in_range(rw) && !S::is_zero(sw) && S::in_range(sw);
    if (reject_high_s) ok = ok && !S::is_high(sw);
    ok = ok && !mp_geq<N>(cx, C::P) && !mp_geq<N>(cy, C::P);
    {
        Affine<C> a;
        a.x = F::from_canonical(cx).e;                   // (values >= p wrap; ok is already false for them)
        a.y = F::from_canonical(cy).e;
        ok = ok && Group<C>::on_curve(a, Group<C>::curve_b());
    }
    uint32_t u1[N], u2[N];
    {
        uint32_t zr[N], w[N];
        S::reduce_wire(zr, zw);
        S::inv(w, sw);
        S::mul(u1, zr, w);
        S::mul(u2, rw, w);
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        u1[j] = ok ? u1[j] : 0u;
        u2[j] = ok ? u2[j] : 0u;
        cx[j] = ok ? cx[j] : C::GX[j];
        cy[j] = ok ? cy[j] : C::GY[j];
    }
    store_wire<C>(u1_out + i * WB, u1);
    store_wire<C>(u2_out + i * WB, u2);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}

// finish: x(R) mod n == r
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_ecdsa_finish(const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf, const uint8_t* __restrict__ r,
               const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    using S = ScalarN<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[N], xr[N], rw[N];
    load_wire<C>(x, r_xy + i * (2 * WB));
    load_wire<C>(rw, r + i * WB);
    S::reduce_once(xr, x);                                // x < p < 2n
    bool eq = true;
#pragma unroll
    for (int j = 0; j < N; j++) eq = eq && (xr[j] == rw[j]);
    ok_out[i] = (valid[i] && !r_inf[i] && eq) ? 1 : 0;
}

// ---- Schnorr (BIP340) verification: k256/src/schnorr/verifying.rs:76-99 ---------------------------------------------
//     e = tagged_hash("BIP0340/challenge", r || pk || m) reduced mod n   (computed by the caller; reduced here)
//     R = s*G + (-e)*P  via mul_by_generator_and_mul_add_vartime;  accept iff R != identity, y(R) even, x(R) == r
// Signature parsing (k256/src/schnorr.rs:132-150): r < p, 0 < s < n.  P is the verifying key's stored affine point
// (lifted with even y by VerifyingKey::from_bytes — ecgpu_batch_decompress with y_is_odd = 0).
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_prepare(const uint8_t* __restrict__ e, const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                  const uint8_t* __restrict__ p_xy, size_t n, uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out,
                  uint8_t* __restrict__ q_out, uint8_t* __restrict__ valid) {
    using S = ScalarN<C>;
    using F = Field<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t ew[N], rw[N], sw[N], cx[N], cy[N];
    load_wire<C>(ew, e + i * WB);
    load_wire<C>(rw, r + i * WB);
    load_wire<C>(sw, s + i * WB);
    load_wire<C>(cx, p_xy + i * (2 * WB));
    load_wire<C>(cy, p_xy + i * (2 * WB) + WB);
    bool ok = !mp_geq<N>(rw, C::P) && !S::is_zero(sw) && S::in_range(sw);
    ok = ok && !mp_geq<N>(cx, C::P) && !mp_geq<N>(cy, C::P);
    {
        Affine<C> a;
        a.x = F::from_canonical(cx).e;
        a.y = F::from_canonical(cy).e;
        ok = ok && Group<C>::on_curve(a, Group<C>::curve_b());
    }
    uint32_t er[N], ne[N];
    S::reduce_once(er, ew);
    {   // -e mod n
        uint32_t d[N];
        bool z = S::is_zero(er);
        mp_sub<N>(d, C::ORDER, er);
#pragma unroll
        for (int j = 0; j < N; j++) ne[j] = z ? 0u : d[j];
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        sw[j] = ok ? sw[j] : 0u;
        ne[j] = ok ? ne[j] : 0u;
        cx[j] = ok ? cx[j] : C::GX[j];
        cy[j] = ok ? cy[j] : C::GY[j];
    }
    store_wire<C>(a_out + i * WB, sw);
    store_wire<C>(b_out + i * WB, ne);
    store_wire<C>(q_out + i * (2 * WB), cx);
    store_wire<C>(q_out + i * (2 * WB) + WB, cy);
    valid[i] = ok ? 1 : 0;
}

// The whole of `VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)` (k256/src/schnorr/verifying.rs:76-99,149-160) from
// wire bytes: lift_x of the 32-byte key (even y; fails for x >= p or a non-residue), signature parsing, the challenge
// hash e = tagged_hash("BIP0340/challenge", r || pk || msg) on the device, then the same a = s, b = -e as above.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_prepare_raw(const uint8_t* __restrict__ pk_x, const uint8_t* __restrict__ msgs, size_t msg_len,
                      const uint8_t* __restrict__ sigs, size_t n, uint8_t* __restrict__ a_out, uint8_t* __restrict__ b_out,
                      uint8_t* __restrict__ q_out, uint8_t* __restrict__ r_out, uint8_t* __restrict__ valid) {
    using S = ScalarN<C>;
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    static_assert(N == 8 && C::A_IS_ZERO, "BIP340 is defined over secp256k1");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rw[N], sw[N], cx[N], cy[N];
    load_be_vec<N>(cx, pk_x + i * 32);
    load_be_vec<N>(rw, sigs + i * 64);
    load_be_vec<N>(sw, sigs + i * 64 + 32);
    bool ok = !mp_geq<N>(rw, C::P) && !S::is_zero(sw) && S::in_range(sw) && !mp_geq<N>(cx, C::P);
    {   // lift_x: y = sqrt(x^3 + 7), the even root
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
    }
    uint32_t ew[N], er[N], ne[N];
    Sha256::bip340_challenge(ew, sigs + i * 64, pk_x + i * 32, msgs + i * msg_len, msg_len);
    S::reduce_once(er, ew);
    {
        uint32_t d[N];
        bool z = S::is_zero(er);
        mp_sub<N>(d, C::ORDER, er);
#pragma unroll
        for (int j = 0; j < N; j++) ne[j] = z ? 0u : d[j];
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        sw[j] = ok ? sw[j] : 0u;
        ne[j] = ok ? ne[j] : 0u;
        cx[j] = ok ? cx[j] : C::GX[j];
        cy[j] = ok ? cy[j] : C::GY[j];
    }
    store_be_vec<N>(a_out + i * 32, sw);
    store_be_vec<N>(b_out + i * 32, ne);
    store_be_vec<N>(q_out + i * 64, cx);
    store_be_vec<N>(q_out + i * 64 + 32, cy);
    store_be_vec<N>(r_out + i * 32, rw);
    valid[i] = ok ? 1 : 0;
}

template <class C>
__global__ void __launch_bounds__(BLOCK)
k_schnorr_finish(const uint8_t* __restrict__ r_xy, const uint8_t* __restrict__ r_inf, const uint8_t* __restrict__ r,
                 const uint8_t* __restrict__ valid, size_t n, uint8_t* __restrict__ ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[N], y[N], rw[N];
    load_wire<C>(x, r_xy + i * (2 * WB));
    load_wire<C>(y, r_xy + i * (2 * WB) + WB);
    load_wire<C>(rw, r + i * WB);
    bool eq = true;
#pragma unroll
    for (int j = 0; j < N; j++) eq = eq && (x[j] == rw[j]);
    ok_out[i] = (valid[i] && !r_inf[i] && !(y[0] & 1u) && eq) ? 1 : 0;
}

// ---- ECDH: x-coordinate of k*P -----------------------------------------------------------------------------------------
// `elliptic_curve::ecdh::diffie_hellman` (elliptic-curve 0.14.1, un-vendored; used through k256/src/ecdh.rs,
// p256/src/ecdh.rs, p384/src/ecdh.rs): SharedSecret = x((public * secret).to_affine()).  The scalar multiplication is
// k_var_base; this kernel keeps x and reports ok = 0 for an identity result (only k = 0 can produce one: the
// reference's NonZeroScalar / PublicKey types exclude it).
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_extract_x(const uint8_t* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n, uint8_t* __restrict__ out_x,
            uint8_t* __restrict__ ok) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[N];
    copy_wire<C>(out_x + i * WB, xy + i * (2 * WB));
    ok[i] = inf[i] ? 0 : 1;
}

// ---- point decompression: DecompressPoint::decompress(x_bytes, y_is_odd) -----------------------------------------------
// primeorder/src/affine.rs:183-200, k256/src/arithmetic/affine.rs:261-280: alpha = x^3 + a x + b,
// beta = sqrt(alpha), y = beta or -beta by the parity of the canonical value.  ok = 0 (zero record) for x >= p or a
// non-residue.
template <class C>
__global__ void __launch_bounds__(BLOCK)
k_decompress(const uint8_t* __restrict__ xs, const uint8_t* __restrict__ y_is_odd, size_t n, uint8_t* __restrict__ out_xy,
             uint8_t* __restrict__ ok_out) {
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N, WB = WireBytes<C>::value;
    (void)N; (void)WB;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t cx[N], cy[N];
    load_wire<C>(cx, xs + i * WB);
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
    if (((cy[0] & 1u) != 0) != (y_is_odd[i] != 0)) {            // the other root: p - beta (beta != 0 here, or parity
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
    store_wire<C>(out_xy + i * (2 * WB), cx);
    store_wire<C>(out_xy + i * (2 * WB) + WB, cy);
    ok_out[i] = ok ? 1 : 0;
}

}  // namespace ecgpu
