// ecgpu_scalar.h — arithmetic modulo the group order n on N saturated 32-bit words (host + device).
//
// Only the ECDSA verification path needs it (three operations per signature: s^-1, z*s^-1, r*s^-1), so this is
// the plain word-serial Montgomery multiplication, not a tuned one.  Reference counterparts: the `Scalar` types
// (k256/src/arithmetic/scalar.rs:100-143 mul / invert, p256/src/arithmetic/scalar.rs, primefield MontyFieldElement
// for p384) and `Reduce<FieldBytes>` (k256 scalar.rs:618-631: one conditional subtraction of n).
#pragma once

#include "ecgpu_modinv.h"
#include "ecgpu_params.h"

namespace ecgpu {

template <class C>
struct ScalarN {
    ECGPU_CONST int N = C::N;

    static ECGPU_HD bool is_zero(const uint32_t* a) { return mp_is_zero<N>(a); }
    static ECGPU_HD bool in_range(const uint32_t* a) { return !mp_geq<N>(a, C::ORDER); }     // a < n
    // a mod n for a < 2^(32 N) < 2n
    static ECGPU_HD void reduce_once(uint32_t* r, const uint32_t* a) {
        uint32_t d[N];
        uint32_t borrow = mp_sub<N>(d, a, C::ORDER);
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = borrow ? a[i] : d[i];
    }
    // a wire-sized value (a message digest, an x coordinate) mod n.  One conditional subtraction where the wire size is
    // the size of n; a 66-byte p521 value can be 2^7 times n (n = 2^521 - 2^260..): up to 128 subtractions there.
    static ECGPU_HD void reduce_wire(uint32_t* r, const uint32_t* a) {
        constexpr int REPS = C::ID == CURVE_P521 ? 128 : 1;
        reduce_once(r, a);
#pragma unroll 1
        for (int i = 1; i < REPS; i++) reduce_once(r, r);
    }
    // a * b * 2^(-32 N) mod n  (CIOS), inputs < n, output < n
    static ECGPU_HD void mont_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
        uint32_t t[N + 2];
#pragma unroll
        for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                c += (uint64_t)a[j] * b[i] + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[N];
            t[N] = (uint32_t)c;
            t[N + 1] = (uint32_t)(c >> 32);
            const uint32_t m = t[0] * C::ORDER_NINV32;
            c = (uint64_t)m * C::ORDER[0] + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < N; j++) {
                c += (uint64_t)m * C::ORDER[j] + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[N];
            t[N - 1] = (uint32_t)c;
            t[N] = t[N + 1] + (uint32_t)(c >> 32);
        }
        uint32_t d[N];
        uint32_t borrow = mp_sub<N>(d, t, C::ORDER);
        const bool use_d = t[N] != 0 || !borrow;
#pragma unroll
        for (int i = 0; i < N; i++) r[i] = use_d ? d[i] : t[i];
    }
    // a * b mod n
    static ECGPU_HD void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
        uint32_t t[N];
        mont_mul(t, a, b);
        mont_mul(r, t, C::ORDER_R2);
    }
    // a R mod n (the Montgomery form of a < n), and back
    static ECGPU_HD void to_mont(uint32_t* r, const uint32_t* a) { mont_mul(r, a, C::ORDER_R2); }
    static ECGPU_HD void from_mont(uint32_t* r, const uint32_t* a) {
        uint32_t one[N];
#pragma unroll
        for (int i = 0; i < N; i++) one[i] = i == 0 ? 1u : 0u;
        mont_mul(r, a, one);
    }
    // 1 / a mod n (0 -> 0), division steps like the reference's Scalar::invert (k256 scalar.rs:139-143)
    static ECGPU_HD void inv(uint32_t* r, const uint32_t* a) { ModInv<N>::invert(r, a, C::ORDER); }
    // a > (n - 1) / 2   (k256 scalar.rs:419-423 IsHigh; ecdsa NORMALIZE_S)
    static ECGPU_HD bool is_high(const uint32_t* a) {
        uint32_t twice[N];
        uint32_t carry = mp_add<N>(twice, a, a);
        return carry || mp_geq<N>(twice, C::ORDER);          // 2a >= n  <=>  a > (n-1)/2 for odd n
    }
};

}  // namespace ecgpu
