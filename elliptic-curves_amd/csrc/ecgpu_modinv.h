// ecgpu_modinv.h — modular inversion by Bernstein–Yang "safegcd" division steps (host + device).
//
// The reference inverts field elements with crypto-bigint's safegcd (k256/src/arithmetic/field.rs:178-184,
// primefield/src/monty.rs:373-375 -> crypto-bigint 0.7 `invert_odd_mod`, un-vendored, Cargo.lock:367-368); the
// inverse is unique, so the method only matters for speed.  On the GPU a Fermat inversion is a serial chain of
// 255 squarings = 45k dependent instructions and dominates k_normalize; the division-step formulation needs
// ~600 cheap 32-bit steps plus 20 small matrix updates, about a quarter of that.
//
// This is the constant-time ("half-delta") variant over signed 30-bit limbs: branch-free, hence lockstep-friendly
// on a 64-wide wavefront.  Published algorithm: Bernstein & Yang, "Fast constant-time gcd computation and modular
// inversion" (2019), in the batched form popularised by libsecp256k1's doc/safegcd_implementation.md:
//   * 30 division steps at a time on the low 30 bits of (f, g) give a 2x2 transition matrix t (entries |.| <= 2^30);
//   * (f, g) <- t (f, g) / 2^30 exactly, and (d, e) <- t (d, e) / 2^30 mod p, made exact by adding the multiple of p
//     that clears the low 30 bits;
//   * after enough steps g = 0, f = +-1 and d = +-x^-1.
// Division-step counts: 518 suffice for 224-bit inputs (18 batches), 590 for 256-bit inputs (20 batches), 886 for 384-bit inputs (31 batches = 930 here,
// extra steps are harmless once g = 0).
#pragma once

#include <cstdint>

#include "ecgpu_params.h"

// -DECGPU_DIVSTEPS_MAD=0: the division steps with masks and additions as until round 5 (A/B: profiles/r06/)
#ifndef ECGPU_DIVSTEPS_MAD
#define ECGPU_DIVSTEPS_MAD 1
#endif

namespace ecgpu {

template <int NW>          // modulus size in 32-bit words: 6, 7, 8, 12 or 17
struct ModInv {
    static_assert(NW == 6 || NW == 7 || NW == 8 || NW == 12 || NW == 17, "192-, 224-, 256-, 384- or 521-bit moduli");
    ECGPU_CONST int NL = NW == 6 ? 7 : NW == 7 ? 8 : NW == 8 ? 9 : NW == 12 ? 13 : 19;   // signed 30-bit limbs (2 spare bits above the modulus)
    // (45907 bits + 26313) / 19929 division steps: 444 / 518 / 591 / 886 / 1255 (521-bit moduli counted as 544 bits)
    ECGPU_CONST int BATCHES = NW == 6 ? 15 : NW == 7 ? 18 : NW == 8 ? 20 : NW == 12 ? 31 : 42;
    ECGPU_CONST int32_t M30 = (int32_t)((1u << 30) - 1);

    struct S30 {
        int32_t v[NL];
    };
    struct Trans {
        int32_t u, v, q, r;
    };

    static ECGPU_HD S30 from_words(const uint32_t* w) {
        S30 r;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            int bit = 30 * l, i = bit / 32, sh = bit % 32;
            uint64_t x = i < NW ? (uint64_t)w[i] >> sh : 0;
            if (i + 1 < NW) x |= (uint64_t)w[i + 1] << (32 - sh);
            r.v[l] = (int32_t)((uint32_t)x & (uint32_t)M30);
        }
        return r;
    }
    static ECGPU_HD void to_words(uint32_t* w, const S30& a) {      // a in [0, p), limbs in [0, 2^30)
#pragma unroll
        for (int i = 0; i < NW; i++) {
            int bit = 32 * i, l = bit / 30, sh = bit % 30;
            uint64_t x = (uint64_t)(uint32_t)a.v[l] >> sh;
            if (l + 1 < NL) x |= (uint64_t)(uint32_t)a.v[l + 1] << (30 - sh);
            if (l + 2 < NL) x |= (uint64_t)(uint32_t)a.v[l + 2] << (60 - sh);
            w[i] = (uint32_t)x;
        }
    }
    // p^-1 mod 2^30 from the low word of p (Newton iteration, exact for odd p)
    static ECGPU_HD uint32_t inv30(uint32_t p0) {
        uint32_t x = p0;                       // correct to 3 bits
        x *= 2 - p0 * x;                       // 6
        x *= 2 - p0 * x;                       // 12
        x *= 2 - p0 * x;                       // 24
        x *= 2 - p0 * x;                       // 48
        return x & (uint32_t)M30;
    }

    // 30 half-delta division steps on the low bits; zeta = -(delta + 1/2)
    //
    // Round 6: the conditional additions of a step as MULTIPLY-ADDS by a multiplier in {-1, 0, +1} resp. {0, 1} — g += (+-f) & c2 is
    // g += f * m1 with m1 = sign(zeta) restricted to "g odd", f += g & c1 is f += g * m2 — on 64-bit accumulators whose upper halves are
    // never read (the product of a 32-bit value by 0xFFFFFFFF is its negative modulo 2^32: only the lower half of a pair carries the
    // value, the carry into the upper half is garbage and stays there).  One v_mad_u64_u32 replaces xor + sub + and + add: 16
    // instructions per step instead of 23 (ISA of k_normalize<K256Params, 0>, tools/isa_loops.py).  A division-step inversion is a lone
    // wave's instruction COUNT wherever it matters (k_normalize, k_msm_combine, k_scalar_batch_inv: one wave per SIMD issues an
    // instruction every ~7 cycles whatever it is), so the six full-rate multiply-adds cost nothing there; where the SIMD is shared
    // (the table build of the ladders) six 4-cycle and ten 2-cycle instructions replace twenty-three 2-cycle ones: equal.
    // Same branch-free schedule, same matrices: the g++ host twin runs this very code (tests/hostcheck).
    // Device: the six multiply-adds and the three shifts of a step are two assembly statements (the compiler, left to itself, proves
    // that only the lower halves are read, takes the pairs apart again and ends up with MORE instructions than the masked form — 142
    // for five steps; and it separates single-instruction asm statements by s_nop).  The shifts take the whole pair along.  Left: the
    // lower half is exact modulo 2^32, which is all u and v are.  Right (g): garbage from the upper half enters at bit 31 and moves
    // down one bit per step — a step only looks at bit 0, and the 30 steps of a batch only depend on the low 30 bits of f and g (the
    // reason a batch can work on one limb at all): it never reaches a bit that is read.
    static ECGPU_HD void step_a(uint64_t& g, uint64_t& q, uint64_t& r, uint32_t fl, uint32_t ul, uint32_t vl, uint32_t m1) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t carry;
        asm("v_mad_u64_u32 %0, %3, %4, %7, %0\n\t"
            "v_mad_u64_u32 %1, %3, %5, %7, %1\n\t"
            "v_mad_u64_u32 %2, %3, %6, %7, %2"
            : "+v"(g), "+v"(q), "+v"(r), "=&s"(carry)
            : "v"(fl), "v"(ul), "v"(vl), "v"(m1));
#else
        g += (uint64_t)fl * m1;
        q += (uint64_t)ul * m1;
        r += (uint64_t)vl * m1;
#endif
    }
    static ECGPU_HD void step_b(uint64_t& f, uint64_t& u, uint64_t& v, uint64_t& g, uint32_t gl, uint32_t ql, uint32_t rl, uint32_t m2) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t carry;
        asm("v_mad_u64_u32 %0, %4, %5, %8, %0\n\t"
            "v_mad_u64_u32 %1, %4, %6, %8, %1\n\t"
            "v_mad_u64_u32 %2, %4, %7, %8, %2\n\t"
            "v_lshrrev_b64 %3, 1, %3\n\t"
            "v_lshlrev_b64 %1, 1, %1\n\t"
            "v_lshlrev_b64 %2, 1, %2"
            : "+v"(f), "+v"(u), "+v"(v), "+v"(g), "=&s"(carry)
            : "v"(gl), "v"(ql), "v"(rl), "v"(m2));
#else
        f += (uint64_t)gl * m2;
        u += (uint64_t)ql * m2;
        v += (uint64_t)rl * m2;
        g >>= 1;
        u <<= 1;
        v <<= 1;
#endif
    }
    static ECGPU_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans* t) {
#if ECGPU_DIVSTEPS_MAD
        uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;           // the values are the LOWER halves
#pragma unroll 5
        for (int i = 0; i < 30; i++) {
            const uint32_t neg = (uint32_t)(zeta >> 31);                   // all ones iff zeta < 0  (delta > 0)
            const uint32_t c2 = 0u - ((uint32_t)g & 1u);                   // all ones iff g odd
            const uint32_t m1 = (neg | 1u) & c2;                           // g odd: -1 if delta > 0 else +1; g even: 0
            step_a(g, q, r, (uint32_t)f, (uint32_t)u, (uint32_t)v, m1);    // g += (+-f) if g odd; the same on the matrix rows
            const uint32_t c1 = neg & c2;                                  // swap iff delta > 0 and g odd
            zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
            step_b(f, u, v, g, (uint32_t)g, (uint32_t)q, (uint32_t)r, c1 & 1u);    // f += g (the new g) if swapping; g >>= 1, u <<= 1, v <<= 1
        }
        t->u = (int32_t)(uint32_t)u;
        t->v = (int32_t)(uint32_t)v;
        t->q = (int32_t)(uint32_t)q;
        t->r = (int32_t)(uint32_t)r;
        return zeta;
#else
        uint32_t u = 1, v = 0, q = 0, r = 1;
        uint32_t f = f0, g = g0;
#pragma unroll 5
        for (int i = 0; i < 30; i++) {
            uint32_t c1 = (uint32_t)(zeta >> 31);          // all ones iff zeta < 0  (delta > 0)
            uint32_t c2 = 0u - (g & 1u);                   // all ones iff g odd
            uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
            g += x & c2;
            q += y & c2;
            r += z & c2;
            c1 &= c2;                                      // swap iff delta > 0 and g odd
            zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
            f += g & c1;
            u += q & c1;
            v += r & c1;
            g >>= 1;
            u <<= 1;
            v <<= 1;
        }
        t->u = (int32_t)u;
        t->v = (int32_t)v;
        t->q = (int32_t)q;
        t->r = (int32_t)r;
        return zeta;
#endif
    }

    // acc + a b on signed 32-bit factors.  On the device as v_mad_i64_i32 by hand: the limbs are known to be non-negative, the
    // compiler therefore rewrites sext(limb) as zext(limb), no longer recognises the signed multiply-add and expands every product
    // into an unsigned multiply-add, a second one for the sign of the other factor and two moves (the round-5 ISA of k_normalize:
    // 118 + 34 multiplies and 119 moves per batch for 90 products).  SA: `a` is wave-uniform (a limb of the constant modulus).
    template <bool SA = false>
    static ECGPU_HD int64_t smad(int32_t a, int32_t b, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
        int64_t r;
        if constexpr (SA) asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(acc) : "vcc");
        else asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
        return r;
#else
        return acc + (int64_t)a * b;
#endif
    }

    // cx += a1 b1 + a2 b2 and cy += a3 b1 + a4 b2: the four products of one limb of the matrix update as ONE assembly statement (the
    // compiler separates single-instruction statements by s_nop: 55 of them per batch of 30 steps until round 6)
    static ECGPU_HD void smad4(int64_t& cx, int64_t& cy, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t b1, int32_t b2) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t carry;
        asm("v_mad_i64_i32 %0, %2, %3, %7, %0\n\t"
            "v_mad_i64_i32 %1, %2, %5, %7, %1\n\t"
            "v_mad_i64_i32 %0, %2, %4, %8, %0\n\t"
            "v_mad_i64_i32 %1, %2, %6, %8, %1"
            : "+v"(cx), "+v"(cy), "=&s"(carry)
            : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2));
#else
        cx += (int64_t)a1 * b1 + (int64_t)a2 * b2;
        cy += (int64_t)a3 * b1 + (int64_t)a4 * b2;
#endif
    }
    // the same + the multiples of the modulus limb pl (wave-uniform: a scalar register): cx += pl mx, cy += pl my
    static ECGPU_HD void smad6(int64_t& cx, int64_t& cy, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t b1, int32_t b2,
                               int32_t pl, int32_t mx, int32_t my) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t carry;
        asm("v_mad_i64_i32 %0, %2, %3, %7, %0\n\t"
            "v_mad_i64_i32 %1, %2, %5, %7, %1\n\t"
            "v_mad_i64_i32 %0, %2, %4, %8, %0\n\t"
            "v_mad_i64_i32 %1, %2, %6, %8, %1\n\t"
            "v_mad_i64_i32 %0, %2, %9, %10, %0\n\t"
            "v_mad_i64_i32 %1, %2, %9, %11, %1"
            : "+v"(cx), "+v"(cy), "=&s"(carry)
            : "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(b1), "v"(b2), "s"(pl), "v"(mx), "v"(my));
#else
        cx += (int64_t)a1 * b1 + (int64_t)a2 * b2 + (int64_t)pl * mx;
        cy += (int64_t)a3 * b1 + (int64_t)a4 * b2 + (int64_t)pl * my;
#endif
    }

    // (d, e) <- t (d, e) / 2^30 mod p, with d, e kept in (-2p, p)
    static ECGPU_HD void update_de(S30& d, S30& e, const Trans& t, const S30& p, uint32_t pinv30) {
        const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
        const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
        int32_t md = (u & sd) + (v & se);
        int32_t me = (q & sd) + (r & se);
        int32_t di = d.v[0], ei = e.v[0];
        int64_t cd = smad(u, di, smad(v, ei, 0));
        int64_t ce = smad(q, di, smad(r, ei, 0));
        md -= (int32_t)((pinv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((pinv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd = smad<true>(p.v[0], md, cd);
        ce = smad<true>(p.v[0], me, ce);
        cd >>= 30;                                         // the low 30 bits are zero by construction
        ce >>= 30;
#pragma unroll
        for (int i = 1; i < NL; i++) {
            di = d.v[i];
            ei = e.v[i];
            smad6(cd, ce, u, v, q, r, di, ei, p.v[i], md, me);
            d.v[i - 1] = (int32_t)cd & M30;
            cd >>= 30;
            e.v[i - 1] = (int32_t)ce & M30;
            ce >>= 30;
        }
        d.v[NL - 1] = (int32_t)cd;
        e.v[NL - 1] = (int32_t)ce;
    }
    // (f, g) <- t (f, g) / 2^30, exact
    static ECGPU_HD void update_fg(S30& f, S30& g, const Trans& t) {
        const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
        int32_t fi = f.v[0], gi = g.v[0];
        int64_t cf = smad(u, fi, smad(v, gi, 0));
        int64_t cg = smad(q, fi, smad(r, gi, 0));
        cf >>= 30;
        cg >>= 30;
#pragma unroll
        for (int i = 1; i < NL; i++) {
            fi = f.v[i];
            gi = g.v[i];
            smad4(cf, cg, u, v, q, r, fi, gi);
            f.v[i - 1] = (int32_t)cf & M30;
            cf >>= 30;
            g.v[i - 1] = (int32_t)cg & M30;
            cg >>= 30;
        }
        f.v[NL - 1] = (int32_t)cf;
        g.v[NL - 1] = (int32_t)cg;
    }
    // d in (-2p, p), sign of f -> sign * d reduced to [0, p) with limbs in [0, 2^30)
    static ECGPU_HD void normalize(S30& d, int32_t fsign, const S30& p) {
        int32_t cond_add = d.v[NL - 1] >> 31;
        const int32_t cond_neg = fsign >> 31;
        int32_t carry = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            int32_t x = d.v[i] + (p.v[i] & cond_add);
            x = (x ^ cond_neg) - cond_neg;
            x += carry;
            if (i + 1 < NL) {
                carry = x >> 30;
                x &= M30;
            }
            d.v[i] = x;
        }
        cond_add = d.v[NL - 1] >> 31;
        carry = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            int32_t x = d.v[i] + (p.v[i] & cond_add) + carry;
            if (i + 1 < NL) {
                carry = x >> 30;
                x &= M30;
            }
            d.v[i] = x;
        }
    }

    // out = x^-1 mod p (0 for x = 0); x canonical (< p), p odd, all as NW little-endian words
    static ECGPU_HD void invert(uint32_t* out, const uint32_t* x, const uint32_t* pw) {
        const S30 p = from_words(pw);
        const uint32_t pinv30 = inv30(pw[0]);
        S30 f = p, g = from_words(x), d, e;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            d.v[i] = 0;
            e.v[i] = i == 0 ? 1 : 0;
        }
        int32_t zeta = -1;
#pragma unroll 1
        for (int it = 0; it < BATCHES; it++) {
            Trans t;
            zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], &t);
            update_de(d, e, t, p, pinv30);
            update_fg(f, g, t);
        }
        normalize(d, f.v[NL - 1], p);
        to_words(out, d);
    }

    // ---- the variable-time form: for ONE public value inverted by a wave (the affine conversion at the end of k_msm_combine) ----
    // Measured (profiles/r05/inversion_ab.txt): the conversion of an MSM's result 0.154 -> 0.144 ms (combine stage, 2^21 terms); in
    // k_normalize, where the 64 lanes of a wave invert 64 different values and every batch takes as long as its slowest lane, the
    // same 0.114 ms as the branch-free form — k_normalize keeps the branch-free one for every caller.
    // The same batches of 30 division steps with the same transition matrices, but the steps of a batch are taken several at a
    // time (Bernstein-Yang's original delta, eta = -delta; libsecp256k1's doc/safegcd_implementation.md, "variable time"): all the
    // trailing zeros of g at once, and then the multiple of f that clears the low min(eta + 1, remaining, 6) bits of g in one
    // addition (w = -g / f mod 2^6 = f g (f^2 - 2), Newton from f^2 = 1 mod 8).  A lane takes 8-10 iterations per batch instead of
    // 30 steps; a wave takes as many as its slowest lane, and stops when g = 0 in all of them (g = 0 is a fixed point: further
    // batches change nothing that is read).  724 steps bound the original delta for 256-bit inputs: at most MAXB batches.
    ECGPU_CONST int MAXB_VAR = NW == 6 ? 19 : NW == 7 ? 22 : NW == 8 ? 25 : NW == 12 ? 37 : 52;     // (49 bits + 57) / 17 steps, / 30

    static ECGPU_HD uint32_t mad_lo(uint32_t a, uint32_t b, uint32_t c) {      // a b + c mod 2^32 (full-rate on the device)
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t r;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"((uint64_t)c) : "vcc");
        return (uint32_t)r;
#else
        return a * b + c;
#endif
    }
    static ECGPU_HD int32_t divsteps_30_var(int32_t eta, uint32_t f0, uint32_t g0, Trans* t) {
        uint32_t u = 1, v = 0, q = 0, r = 1;
        uint32_t f = f0, g = g0;
        int i = 30;
        for (;;) {
            const int zeros = __builtin_ctz(g | (0xFFFFFFFFu << i));         // (a sentinel bit: never more than i)
            g >>= zeros;
            u <<= zeros;
            v <<= zeros;
            eta -= zeros;
            i -= zeros;
            if (i == 0) break;
            if (eta < 0) {                                                   // delta > 0 and g odd: (f, g) <- (g, -f)
                uint32_t tmp;
                eta = -eta;
                tmp = f; f = g; g = 0u - tmp;
                tmp = u; u = q; q = 0u - tmp;
                tmp = v; v = r; r = 0u - tmp;
            }
            const int limit = (eta + 1) > i ? i : (eta + 1);
            const uint32_t m = (0xFFFFFFFFu >> (32 - limit)) & 63u;
            const uint32_t fl = f & 63u, gl = g & 63u;
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t w = __umul24(__umul24(fl, gl), __umul24(fl, fl) - 2u) & m;    // -g / f mod 2^min(limit, 6) (24-bit multiplies: full rate)
#else
            const uint32_t w = (fl * gl * (fl * fl - 2u)) & m;
#endif
            g = mad_lo(f, w, g);
            q = mad_lo(u, w, q);
            r = mad_lo(v, w, r);
        }
        t->u = (int32_t)u;
        t->v = (int32_t)v;
        t->q = (int32_t)q;
        t->r = (int32_t)r;
        return eta;
    }
    static ECGPU_HD void invert_var(uint32_t* out, const uint32_t* x, const uint32_t* pw) {
        const S30 p = from_words(pw);
        const uint32_t pinv30 = inv30(pw[0]);
        S30 f = p, g = from_words(x), d, e;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            d.v[i] = 0;
            e.v[i] = i == 0 ? 1 : 0;
        }
        int32_t eta = -1;
#pragma unroll 1
        for (int it = 0; it < MAXB_VAR; it++) {
            Trans t;
            eta = divsteps_30_var(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], &t);
            update_de(d, e, t, p, pinv30);
            update_fg(f, g, t);
            int32_t any = 0;
#pragma unroll
            for (int i = 0; i < NL; i++) any |= g.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
            if (__builtin_amdgcn_ballot_w64(any != 0) == 0) break;           // wave-uniform: every lane is done
#else
            if (any == 0) break;
#endif
        }
        normalize(d, f.v[NL - 1], p);
        to_words(out, d);
    }
};

}  // namespace ecgpu
