// ecgpu_point.h — short-Weierstrass group law on homogeneous projective coordinates with the
// Renes–Costello–Batina complete formulas, exactly the coordinate system and formulas the reference
// uses (so there are no exceptional cases to special-case on the GPU either):
//
//   a = 0  (k256)       k256/src/arithmetic/projective.rs:96-131 (Alg 7 add), :142-176 (Alg 8 mixed
//                       add), :189-217 (Alg 9 double); b3 = 3*7 = 21 as a small constant
//   a = -3 (p256/p384)  primeorder/src/point_arithmetic.rs:222-245 (Alg 4), :254-280 (Alg 5),
//                       :289-318 (Alg 6)
//   identity            (0 : 1 : 0)   k256 projective.rs:49-53, primeorder projective.rs:60-64
//
// The formulas are written once against the magnitude-typed field interface (ecgpu_field.h): every
// intermediate carries its limb/value bounds in its type, `norm` is inserted exactly where the lazily
// reduced representations need it, and the sums of two
// products that end every formula are evaluated with ONE reduction (`mul2`).  This is the same
// bookkeeping the reference does by hand with `negate(m)` / `normalize_weak()` in
// k256/src/arithmetic/projective.rs.
//
// Coordinates of stored points (Proj / Affine) always have magnitude (1, 1).
#pragma once

#include <type_traits>

#include "ecgpu_field.h"

namespace ecgpu {

template <class C>
struct Affine {  // identity is NOT representable here; callers carry a flag / skip
    Fe<C::NL> x, y;
};

template <int N2>
struct PackedPoint {   // affine point in packed storage form (2 x N fully reduced words), as gathered from HBM
    uint32_t w[N2];
};

template <class C>
struct Proj {
    Fe<C::NL> x, y, z;
};

// Jacobian coordinates (x = X/Z^2, y = Y/Z^3) for the variable-base ladder, where the accumulator is provably
// never the identity and never +-(the table entry) before the last digit (see ecgpu_varmul.h); the identity is
// NOT representable.  Coordinates have limb magnitude 1 and value magnitude <= Group::JV.
template <class C>
struct Jac {
    Fe<C::NL> x, y, z;
};

// Extended Jacobian ("XYZZ") coordinates: x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2.  The accumulator of sums of affine
// points (fixed-base comb): a mixed addition is 8M + 2S with nine reductions, against 8M + 3S / ten for Jacobian and
// 11M for the complete formulas.  The identity is not representable.
template <class C>
struct Xyzz {
    Fe<C::NL> x, y, zz, zzz;
};

template <class C, int L, int V>
std::integral_constant<int, V> magv(const Mag<C, L, V>&);


template <class C>
struct Group {
    using F = Field<C>;
    using E = Fe<C::NL>;
    using M1 = typename F::M1;
    using P = Proj<C>;
    using A = Affine<C>;

    static ECGPU_HD M1 m(const E& e) { return F::template wrap<1, 1>(e); }

    static ECGPU_HD P identity() {
        P r;
        r.x = F::zero().e;
        r.y = F::one().e;
        r.z = F::zero().e;
        return r;
    }
    static ECGPU_HD bool is_identity(const P& p) { return F::is_zero(m(p.z)); }

    static ECGPU_HD P from_affine(const A& a) {
        P r;
        r.x = a.x;
        r.y = a.y;
        r.z = F::one().e;
        return r;
    }
    // -y as a stored coordinate of magnitude (1, 1).  For the Montgomery-lazy field the value magnitude is
    // brought back to 1 by a multiplication with the Montgomery one; the hot loops never call this — they
    // pass the sign into add / add_mixed instead.
    static ECGPU_HD E neg_coord(const E& y) {
        if constexpr (C::REPR == REPR_U28_MONT) return F::mul(F::neg(m(y)), F::one()).e;
        else return F::norm(F::neg(m(y))).e;
    }
    static ECGPU_HD P neg(const P& p) {
        P r = p;
        r.y = neg_coord(p.y);
        return r;
    }
    static ECGPU_HD A neg(const A& p) {
        A r = p;
        r.y = neg_coord(p.y);
        return r;
    }
    static ECGPU_HD E curve_b() {  // curve b in internal form (a = -3 curves); unused for k256
        if constexpr (C::REPR == REPR_U28_MONT) {
            return F::p_const(C::UC::BM);
        } else {
            E b = F::zero().e;
            b.v[0] = C::B_SMALL;
            return b;
        }
    }

    // ---- a = 0 (k256) ------------------------------------------------------------------------
    // b3 = 21.  Magnitudes in comments are limb magnitudes; the product limit is 7.
    // negq: add -q instead of q (the sign is folded into Y2, no separate negation pass)
    static ECGPU_HD P add_a0(const P& p, const P& q, bool negq) {
        constexpr uint32_t b3 = 3 * C::B_SMALL;
        auto X1 = m(p.x), Y1 = m(p.y), Z1 = m(p.z), X2 = m(q.x), Z2 = m(q.z);
        auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));       // 2
        auto xx = F::mul(X1, X2);
        auto yy = F::mul(Y1, Y2);
        auto zz = F::mul(Z1, Z2);
        auto xy = F::mul_sub(F::add(X1, Y1), F::add(X2, Y2), F::add(xx, yy));   // 4 -> 1
        auto yz = F::mul_sub(F::add(Y1, Z1), F::add(Y2, Z2), F::add(yy, zz));
        auto xz = F::mul_sub(F::add(X1, Z1), F::add(X2, Z2), F::add(xx, zz));
        auto bzz3 = F::template mul_small<b3>(zz);
        auto yy_m = F::sub(yy, bzz3);                       // 3
        auto yy_p = F::add(yy, bzz3);                       // 2
        auto byz3 = F::template mul_small<b3>(yz);
        auto xx3 = F::add(F::dbl(xx), xx);                  // 3
        auto bxx9 = F::template mul_small<3 * b3>(xx);
        P r;
        r.x = F::mul2(xy, yy_m, F::neg(byz3), xz).e;        // 1*3 + 2*1 = 5
        r.y = F::mul2(yy_p, yy_m, bxx9, xz).e;              // 2*3 + 1*1 = 7
        r.z = F::mul2(yz, yy_p, xx3, xy).e;                 // 1*2 + 3*1 = 5
        return r;
    }
    static ECGPU_HD P add_mixed_a0(const P& p, const A& q, bool negq) {
        constexpr uint32_t b3 = 3 * C::B_SMALL;
        auto X1 = m(p.x), Y1 = m(p.y), Z1 = m(p.z), X2 = m(q.x);
        auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));       // 2
        auto xx = F::mul(X1, X2);
        auto yy = F::mul(Y1, Y2);
        auto xy = F::mul_sub(F::add(X1, Y1), F::add(X2, Y2), F::add(xx, yy));   // 4 -> 1
        auto yz = F::add(F::mul(Y2, Z1), Y1);               // 2
        auto xz = F::add(F::mul(X2, Z1), X1);               // 2
        auto bzz3 = F::template mul_small<b3>(Z1);
        auto yy_m = F::sub(yy, bzz3);                       // 3
        auto yy_p = F::add(yy, bzz3);                       // 2
        auto byz3 = F::template mul_small<b3>(yz);
        auto xx3 = F::add(F::dbl(xx), xx);                  // 3
        auto bxx9 = F::template mul_small<3 * b3>(xx);
        auto yy_mn = F::norm(yy_m);                         // 1
        P r;
        r.x = F::mul2(xy, yy_m, F::neg(byz3), xz).e;        // 1*3 + 2*2 = 7
        r.y = F::mul2(yy_p, yy_mn, bxx9, xz).e;             // 2*1 + 1*2 = 4
        r.z = F::mul2(yz, yy_p, xx3, xy).e;                 // 2*2 + 3*1 = 7
        return r;
    }
    static ECGPU_HD P dbl_a0(const P& p) {
        constexpr uint32_t b3 = 3 * C::B_SMALL;
        auto X = m(p.x), Y = m(p.y), Z = m(p.z);
        auto yy = F::sqr(Y);
        auto zz = F::sqr(Z);
        auto xy2 = F::dbl(F::mul(X, Y));                    // 2
        auto bzz3 = F::template mul_small<b3>(zz);
        auto bzz9 = F::template mul_small<3 * b3>(zz);
        auto yy_m9 = F::sub(yy, bzz9);                      // 3
        auto yy_p3 = F::add(yy, bzz3);                      // 2
        auto yy24b = F::template mul_small<8 * b3>(yy);     // 24 b yy
        auto z8 = F::template mul_small<8>(Z);
        P r;
        r.x = F::mul(xy2, yy_m9).e;                         // 2*3 = 6
        r.y = F::mul2(yy_m9, yy_p3, yy24b, zz).e;           // 3*2 + 1*1 = 7
        r.z = F::mul(F::mul(yy, Y), z8).e;
        return r;
    }

    // ---- a = -3 (p256: product limit 23, magnitude limit 15; p384: product limit 30, magnitude limit 28) ----
    static ECGPU_HD P add_am3(const P& l, const P& r, const E& be, bool negq) {
        auto b = m(be);
        auto X1 = m(l.x), Y1 = m(l.y), Z1 = m(l.z), X2 = m(r.x), Z2 = m(r.z);
        auto Y2 = F::sel(negq, F::neg(m(r.y)), m(r.y));
        auto xx = F::mul(X1, X2);
        auto yy = F::mul(Y1, Y2);
        auto zz = F::mul(Z1, Z2);
        auto xy = F::mul_sub(F::add(X1, Y1), F::add(X2, Y2), F::add(xx, yy));   // 4 -> 1
        auto yz = F::mul_sub(F::add(Y1, Z1), F::add(Y2, Z2), F::add(yy, zz));   // 4 -> 1
        auto xz = F::sub(F::mul(F::add(X1, Z1), F::add(X2, Z2)), F::add(xx, zz));            // 4
        auto bzz = F::norm(F::sub(xz, F::mul(b, zz)));      // 6 -> 1
        auto bzz3 = F::add(F::dbl(bzz), bzz);               // 3
        auto yy_m = F::sub(yy, bzz3);                       // 5
        auto yy_p = F::add(yy, bzz3);                       // 4
        auto zz3 = F::add(F::dbl(zz), zz);                  // 3
        auto bxz = F::mul_sub(b, xz, F::add(zz3, xx));                           // 6 -> 1
        auto bxz3 = F::add(F::dbl(bxz), bxz);               // 3
        auto xx3_m_zz3 = F::norm(F::sub(F::add(F::dbl(xx), xx), zz3));                        // 7 -> 1
        P o;
        o.x = F::mul2(yy_p, xy, F::neg(yz), bxz3).e;        // 4*1 + 2*3 = 10
        o.y = F::mul2(yy_p, yy_m, xx3_m_zz3, bxz3).e;       // 4*5 + 1*3 = 23
        o.z = F::mul2(yy_m, yz, xy, xx3_m_zz3).e;           // 5*1 + 1*1 = 6
        return o;
    }
    static ECGPU_HD P add_mixed_am3(const P& l, const A& r, const E& be, bool negq) {
        auto b = m(be);
        auto X1 = m(l.x), Y1 = m(l.y), Z1 = m(l.z), X2 = m(r.x);
        auto Y2 = F::sel(negq, F::neg(m(r.y)), m(r.y));
        auto xx = F::mul(X1, X2);
        auto yy = F::mul(Y1, Y2);
        auto xy = F::mul_sub(F::add(X1, Y1), F::add(X2, Y2), F::add(xx, yy));   // 4 -> 1
        auto yz = F::add(F::mul(Y2, Z1), Y1);               // 2
        auto xz = F::add(F::mul(X2, Z1), X1);               // 2
        auto bz = F::norm(F::sub(xz, F::mul(b, Z1)));       // 4 -> 1
        auto bz3 = F::add(F::dbl(bz), bz);                  // 3
        auto yy_m = F::sub(yy, bz3);                        // 5
        auto yy_p = F::add(yy, bz3);                        // 4
        auto z3 = F::add(F::dbl(Z1), Z1);                   // 3
        auto bxz = F::mul_sub(b, xz, F::add(z3, xx));                            // 6 -> 1
        auto bxz3 = F::add(F::dbl(bxz), bxz);               // 3
        auto xx3_m_zz3 = F::norm(F::sub(F::add(F::dbl(xx), xx), z3));                         // 7 -> 1
        P o;
        o.x = F::mul2(yy_p, xy, F::neg(yz), bxz3).e;        // 4*1 + 3*3 = 13
        o.y = F::mul2(yy_p, yy_m, xx3_m_zz3, bxz3).e;       // 4*5 + 1*3 = 23
        o.z = F::mul2(yy_m, yz, xy, xx3_m_zz3).e;           // 5*2 + 1*1 = 11
        return o;
    }
    static ECGPU_HD P dbl_am3(const P& p, const E& be) {
        auto b = m(be);
        auto X = m(p.x), Y = m(p.y), Z = m(p.z);
        auto xx = F::sqr(X);
        auto yy = F::sqr(Y);
        auto zz = F::sqr(Z);
        auto xy2 = F::dbl(F::mul(X, Y));                    // 2
        auto xz2 = F::dbl(F::mul(X, Z));                    // 2
        auto bzz = F::mul_sub(b, zz, xz2);     // 4 -> 1
        auto bzz3 = F::add(F::dbl(bzz), bzz);               // 3
        auto yy_m = F::sub(yy, bzz3);                       // 5
        auto yy_p = F::add(yy, bzz3);                       // 4
        auto zz3 = F::add(F::dbl(zz), zz);                  // 3
        auto bxz2 = F::mul_sub(b, xz2, F::add(zz3, xx));                         // 6 -> 1
        auto bxz6 = F::add(F::dbl(bxz2), bxz2);             // 3
        auto xx3_m_zz3 = F::norm(F::sub(F::add(F::dbl(xx), xx), zz3));                        // 7 -> 1
        auto yz2 = F::dbl(F::mul(Y, Z));                    // 2
        P o;
        o.x = F::mul2(yy_m, xy2, F::neg(bxz6), yz2).e;      // 5*2 + 4*2 = 18
        o.y = F::mul2(yy_p, yy_m, xx3_m_zz3, bxz6).e;       // 4*5 + 1*3 = 23
        o.z = F::mul(yz2, F::dbl(F::dbl(yy))).e;            // 2*4 = 8
        return o;
    }

    // ---- Jacobian ladder arithmetic (incomplete formulas; preconditions in ecgpu_varmul.h) -----------
    // The reference's variable-base path uses the complete projective formulas throughout
    // (primeorder/src/projective.rs:532-557); here the 4 doublings per digit — 80% of the work — use the
    // cheaper Jacobian doubling (a = -3: 3M + 5S, a = 0: 2M + 5S) and only the final addition, the one place
    // an exceptional case can occur, is the complete one.
    // value magnitudes: ladder outputs (and so table entries) <= JTV, accumulator <= JV (a negated table entry)
    ECGPU_CONST int JTV = C::REPR == REPR_U28_MONT ? 10 : 1;
    ECGPU_CONST int JV = C::REPR == REPR_U28_MONT ? JTV + 1 : 1;
    using J = Jac<C>;
    static ECGPU_HD Mag<C, 1, JV> mj(const E& e) { return F::template wrap<1, JV>(e); }
    static ECGPU_HD Mag<C, 1, JTV> mt(const E& e) { return F::template wrap<1, JTV>(e); }
    template <int L, int V>
    static ECGPU_HD E jstore(const Mag<C, L, V>& a) {
        static_assert(L == 1 && V <= JTV, "Jacobian coordinate magnitude");
        return a.e;
    }
    static ECGPU_HD J jac_from_affine(const A& a) {
        J r;
        r.x = a.x;
        r.y = a.y;
        r.z = F::one().e;
        return r;
    }
    // value-magnitude-1 copy of a Jacobian coordinate (a no-op where JV = 1)
    static ECGPU_HD E j_unit(const E& c) {
        if constexpr (JV == 1) return c;
        else return F::mul(mj(c), F::one()).e;
    }
    // (X : Y : Z) Jacobian -> (X Z : Y : Z^3) homogeneous
    static ECGPU_HD P jac_to_proj(const J& p) {
        auto X = mj(p.x), Z = mj(p.z);
        P r;
        r.x = F::mul(X, Z).e;
        r.y = j_unit(p.y);
        r.z = F::mul(Z, F::sqr(Z)).e;
        return r;
    }
    // dbl-2001-b (a = -3) / dbl-2009-l (a = 0); valid for every finite point of odd order
    static ECGPU_HD J jac_dbl(const J& p) {
        auto X = mj(p.x), Y = mj(p.y), Z = mj(p.z);
        J o;
        if constexpr (C::A_IS_ZERO) {
            static_assert(C::REPR == REPR_U29_K256, "the a = 0 doubling uses the k256 field's fused subtractions");
            auto aa = F::sqr(X);
            auto bb = F::sqr(Y);
            auto cc = F::sqr(bb);
            auto e3 = F::template mul_small<3>(aa);
            auto d = F::dbl(F::sqr_sub(F::add(X, bb), F::add(aa, cc)));                   // 2   (differences: F::sqr_sub / mul_sub,
            auto X3 = F::sqr_sub(e3, F::dbl(d));                                          //      one reduction each, no norm)
            o.x = jstore(X3);
            o.y = jstore(F::mul_sub(e3, F::sub(d, X3), F::template mul_small<8>(cc)));
            o.z = jstore(F::mul(F::dbl(Y), Z));
        } else {
            auto delta = F::sqr(Z);
            auto gamma = F::sqr(Y);
            auto beta = F::mul(X, gamma);
            // a = -3: 3 (X - delta)(X + delta) = 3 X^2 - 3 delta^2; any a (dbl-2007-bl): 3 X^2 + a delta^2
            auto alpha3 = [&] {
                if constexpr (GenericA<C>::value) {
                    auto xx = F::sqr(X);
                    return F::norm(F::add(F::add(F::dbl(xx), xx), F::mul(curve_a(), F::sqr(delta))));
                } else {
                    auto alpha = F::mul(F::sub(X, delta), F::add(X, delta));              // 3 * 2
                    return F::add(F::dbl(alpha), alpha);                                  // 3
                }
            }();
            auto beta4 = F::dbl(F::dbl(beta));                                            // 4
            auto X3 = F::sqr_sub(alpha3, F::dbl(beta4));                     // 10 -> 1
            auto gg8 = F::dbl(F::dbl(F::dbl(F::sqr(gamma))));                             // 8
            o.x = jstore(X3);
            o.y = jstore(F::mul_sub(alpha3, F::sub(beta4, X3), gg8));        // 3 * 6; 10 -> 1
            o.z = jstore(F::sqr_sub(F::add(Y, Z), F::add(gamma, delta)));    // 4 -> 1
        }
        return o;
    }
    // madd-2004-hmv (Z2 = 1): 8M + 3S.  Requires p finite, p != +-q.  Z3 = Z1 * H; H is handed back for the callers
    // that track Z ratios (the shared-Z table of the k256 ladder).
    static ECGPU_HD J jac_madd(const J& p, const A& q, bool negq, E* h_out = nullptr) {
        auto X1 = mj(p.x), Y1 = mj(p.y), Z1 = mj(p.z);
        auto zz1 = F::sqr(Z1);
        if constexpr (C::REPR == REPR_U29_K256) {         // (as xyzz_madd: the three differences out of their products' reductions)
            auto H = F::mul_sub(m(q.x), zz1, X1);
            if (h_out) *h_out = H.e;
            auto t = F::mul(Z1, zz1);
            J o;
            o.z = jstore(F::mul(Z1, H));
            auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));
            auto r = F::mul_sub(Y2, t, Y1);
            auto HH = F::sqr(H);
            auto V = F::mul(X1, HH);
            auto HHH = F::mul(H, HH);
            auto X3 = F::sqr_sub(r, F::add(HHH, F::dbl(V)));
            o.x = jstore(X3);
            o.y = jstore(F::mul2(r, F::sub(V, X3), F::neg(Y1), HHH));
            return o;
        }
        auto Hn = F::mul_sub(m(q.x), zz1, X1);
        if (h_out) *h_out = Hn.e;
        auto H = F::template fit<F::SQLIM>(Hn);
        auto t = F::mul(Z1, zz1);
        J o;
        o.z = jstore(F::mul(Z1, H));
        auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));
        auto r = F::template fit<F::SQLIM>(F::sub(F::mul(Y2, t), Y1));
        auto HH = F::sqr(H);
        auto V = F::mul(X1, HH);
        auto HHH = F::mul(H, HH);
        auto X3 = F::sqr_sub(r, F::add(HHH, F::dbl(V)));
        o.x = jstore(X3);
        o.y = jstore(F::mul2(r, F::sub(V, X3), F::neg(Y1), HHH));
        return o;
    }

    // ---- XYZZ accumulator for sums of affine points (incomplete; preconditions in ecgpu_fixedmul.h) -------------
    using XZ = Xyzz<C>;
    static ECGPU_HD XZ xyzz_from_affine(const A& a, bool negate) {
        XZ r;
        r.x = a.x;
        r.y = F::sel(negate, F::norm(F::neg(m(a.y))), m(a.y)).e;
        r.zz = F::one().e;
        r.zzz = F::one().e;
        return r;
    }
    // madd-2008-s: 8M + 2S.  Requires p != +-q.  negq adds -q.
    static ECGPU_HD XZ xyzz_madd(const XZ& p, const A& q, bool negq) {
        auto X1 = mj(p.x), Y1 = mj(p.y), ZZ1 = mj(p.zz), ZZZ1 = mj(p.zzz);
        auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));
        if constexpr (C::REPR == REPR_U29_K256) {
            // the three differences that feed a multiplication come out of the reduction of the product they follow (F::mul_sub /
            // F::sqr_sub): no limb-wise subtraction, no carry pass of their own
            auto Pd = F::mul_sub(m(q.x), ZZ1, X1);
            auto R = F::mul_sub(Y2, ZZZ1, Y1);
            auto PP = F::sqr(Pd);
            auto PPP = F::mul(Pd, PP);
            auto Q = F::mul(X1, PP);
            auto X3 = F::sqr_sub(R, F::add(PPP, F::dbl(Q)));
            XZ o;
            o.x = jstore(X3);
            o.y = jstore(F::mul2(R, F::sub(Q, X3), F::neg(Y1), PPP));
            o.zz = jstore(F::mul(ZZ1, PP));
            o.zzz = jstore(F::mul(ZZZ1, PPP));
            return o;
        }
        auto Pd = F::template fit<F::SQLIM>(F::sub(F::mul(m(q.x), ZZ1), X1));
        auto R = F::template fit<F::SQLIM>(F::sub(F::mul(Y2, ZZZ1), Y1));
        auto PP = F::sqr(Pd);
        auto PPP = F::mul(Pd, PP);
        auto Q = F::mul(X1, PP);
        auto X3 = F::sqr_sub(R, F::add(PPP, F::dbl(Q)));
        XZ o;
        o.x = jstore(X3);
        o.y = jstore(F::mul2(R, F::sub(Q, X3), F::neg(Y1), PPP));
        o.zz = jstore(F::mul(ZZ1, PP));
        o.zzz = jstore(F::mul(ZZZ1, PPP));
        return o;
    }
    // mmadd-2008-s (both affine): 4M + 2S.  Requires p != +-q.
    static ECGPU_HD XZ xyzz_mmadd(const A& p, const A& q, bool negq) {
        auto X1 = m(p.x), Y1 = m(p.y);
        auto Y2 = F::sel(negq, F::neg(m(q.y)), m(q.y));
        auto Pd = F::template fit<F::SQLIM>(F::sub(m(q.x), X1));
        auto R = F::template fit<F::SQLIM>(F::sub(Y2, Y1));
        auto PP = F::sqr(Pd);
        auto PPP = F::mul(Pd, PP);
        auto Q = F::mul(X1, PP);
        auto X3 = F::sqr_sub(R, F::add(PPP, F::dbl(Q)));
        XZ o;
        o.x = jstore(X3);
        o.y = jstore(F::mul2(R, F::sub(Q, X3), F::neg(Y1), PPP));
        o.zz = jstore(PP);
        o.zzz = jstore(PPP);
        return o;
    }
    // (X : Y : ZZ : ZZZ) -> (X ZZZ : Y ZZ : ZZ ZZZ) homogeneous
    static ECGPU_HD P xyzz_to_proj(const XZ& p) {
        auto ZZ = mj(p.zz), ZZZ = mj(p.zzz);
        P r;
        r.x = F::mul(mj(p.x), ZZZ).e;
        r.y = F::mul(mj(p.y), ZZ).e;
        r.z = F::mul(ZZ, ZZZ).e;
        return r;
    }

    // ---- generic a (brainpool): RCB Alg 1-3, primeorder/src/point_arithmetic.rs:56-208 ---------------------------
    // Written for correctness, not tuned: every sum is normalised before it is multiplied, the pairs of products that
    // are added go through mul2.  a and 3b are field constants in internal form.
    static ECGPU_HD M1 curve_a() {
        if constexpr (GenericA<C>::value) return m(F::p_const(C::UC::AM));
        else return F::zero();
    }
    static ECGPU_HD P add_gen(const P& l, const P& r, const E& be, bool negq) {
        auto a = curve_a();
        auto b3 = F::norm(F::add(F::dbl(m(be)), m(be)));
        auto X1 = m(l.x), Y1 = m(l.y), Z1 = m(l.z), X2 = m(r.x), Z2 = m(r.z);
        auto Y2 = F::norm(F::sel(negq, F::neg(m(r.y)), F::add(m(r.y), F::zero())));
        auto t0 = F::mul(X1, X2);                                                             // 1
        auto t1 = F::mul(Y1, Y2);                                                             // 2
        auto t2 = F::mul(Z1, Z2);                                                             // 3
        auto t3 = F::mul_sub(F::add(X1, Y1), F::add(X2, Y2), F::add(t0, t1));   // 4-8
        auto t4 = F::mul_sub(F::add(X1, Z1), F::add(X2, Z2), F::add(t0, t2));   // 9-13
        auto t5 = F::mul_sub(F::add(Y1, Z1), F::add(Y2, Z2), F::add(t1, t2));   // 14-18
        auto z3 = F::mul2(a, t4, b3, t2);                                                     // 19-21
        auto x3 = F::norm(F::sub(t1, z3));                                                    // 22
        auto z3p = F::norm(F::add(t1, z3));                                                   // 23
        auto at2 = F::mul(a, t2);                                                             // 27
        auto t1n = F::norm(F::add(F::add(F::dbl(t0), t0), at2));                              // 25, 26, 29
        auto t2n = F::mul(a, F::norm(F::sub(t0, at2)));                                       // 30, 31
        auto t4n = F::norm(F::add(F::mul(b3, t4), t2n));                                      // 28, 32
        P o;
        o.y = F::mul2(x3, z3p, t1n, t4n).e;                                                   // 24, 33, 34
        o.x = F::mul2(t3, x3, F::neg(t5), t4n).e;                                             // 35-37
        o.z = F::mul2(t5, z3p, t3, t1n).e;                                                    // 38-40
        return o;
    }
    static ECGPU_HD P add_mixed_gen(const P& l, const A& r, const E& be, bool negq) {
        auto a = curve_a();
        auto b3 = F::norm(F::add(F::dbl(m(be)), m(be)));
        auto X1 = m(l.x), Y1 = m(l.y), Z1 = m(l.z), X2 = m(r.x);
        auto Y2 = F::norm(F::sel(negq, F::neg(m(r.y)), F::add(m(r.y), F::zero())));
        auto t0 = F::mul(X1, X2);                                                             // 1
        auto t1 = F::mul(Y1, Y2);                                                             // 2
        auto t3 = F::mul_sub(F::add(X2, Y2), F::add(X1, Y1), F::add(t0, t1));   // 3-7
        auto t4 = F::norm(F::add(F::mul(X2, Z1), X1));                                        // 8, 9
        auto t5 = F::norm(F::add(F::mul(Y2, Z1), Y1));                                        // 10, 11
        auto z3 = F::mul2(a, t4, b3, Z1);                                                     // 12-14
        auto x3 = F::norm(F::sub(t1, z3));                                                    // 15
        auto z3p = F::norm(F::add(t1, z3));                                                   // 16
        auto az = F::mul(a, Z1);                                                              // 20
        auto t1n = F::norm(F::add(F::add(F::dbl(t0), t0), az));                               // 18, 19, 22
        auto t2n = F::mul(a, F::norm(F::sub(t0, az)));                                        // 23, 24
        auto t4n = F::norm(F::add(F::mul(b3, t4), t2n));                                      // 21, 25
        P o;
        o.y = F::mul2(x3, z3p, t1n, t4n).e;                                                   // 17, 26, 27
        o.x = F::mul2(t3, x3, F::neg(t5), t4n).e;                                             // 28-30
        o.z = F::mul2(t5, z3p, t3, t1n).e;                                                    // 31-33
        return o;
    }
    static ECGPU_HD P dbl_gen(const P& p, const E& be) {
        auto a = curve_a();
        auto b3 = F::norm(F::add(F::dbl(m(be)), m(be)));
        auto X = m(p.x), Y = m(p.y), Z = m(p.z);
        auto t0 = F::sqr(X);                                                                  // 1
        auto t1 = F::sqr(Y);                                                                  // 2
        auto t2 = F::sqr(Z);                                                                  // 3
        auto t3 = F::norm(F::dbl(F::mul(X, Y)));                                              // 4, 5
        auto z3 = F::norm(F::dbl(F::mul(X, Z)));                                              // 6, 7
        auto y3 = F::mul2(a, z3, b3, t2);                                                     // 8-10
        auto x3 = F::norm(F::sub(t1, y3));                                                    // 11
        auto y3p = F::norm(F::add(t1, y3));                                                   // 12
        auto at2 = F::mul(a, t2);                                                             // 16
        auto t3n = F::norm(F::add(F::mul(a, F::norm(F::sub(t0, at2))), F::mul(b3, z3)));      // 15, 17-19
        auto t0n = F::norm(F::add(F::add(F::dbl(t0), t0), at2));                              // 20-22
        auto t2y = F::norm(F::dbl(F::mul(Y, Z)));                                             // 25, 26
        P o;
        o.y = F::mul2(x3, y3p, t0n, t3n).e;                                                   // 13, 23, 24
        o.x = F::mul2(t3, x3, F::neg(t2y), t3n).e;                                            // 14, 27, 28
        o.z = F::mul(t2y, F::dbl(F::dbl(t1))).e;                                              // 29-31
        return o;
    }

    // ---- curve-generic entry points (b is ignored for a = 0) -----------------------------------
    static ECGPU_HD P add(const P& p, const P& q, const E& b, bool negq = false) {
        if constexpr (C::A_IS_ZERO) return add_a0(p, q, negq);
        else if constexpr (GenericA<C>::value) return add_gen(p, q, b, negq);
        else return add_am3(p, q, b, negq);
    }
    static ECGPU_HD P add_mixed(const P& p, const A& q, const E& b, bool negq = false) {
        if constexpr (C::A_IS_ZERO) return add_mixed_a0(p, q, negq);
        else if constexpr (GenericA<C>::value) return add_mixed_gen(p, q, b, negq);
        else return add_mixed_am3(p, q, b, negq);
    }
    static ECGPU_HD P dbl(const P& p, const E& b) {
        if constexpr (C::A_IS_ZERO) return dbl_a0(p);
        else if constexpr (GenericA<C>::value) return dbl_gen(p, b);
        else return dbl_am3(p, b);
    }

    // y^2 == x^3 + a x + b   (primeorder/src/affine.rs:100-109)
    static ECGPU_HD bool on_curve(const A& p, const E& be) {
        auto x = m(p.x), y = m(p.y);
        auto lhs = F::sqr(y);
        auto x3 = F::mul(F::sqr(x), x);
        if constexpr (C::A_IS_ZERO) {
            return F::eq(lhs, F::add(x3, m(be)));
        } else if constexpr (GenericA<C>::value) {
            return F::eq(lhs, F::add(F::add(x3, F::mul(curve_a(), x)), m(be)));
        } else {
            auto x3x = F::add(F::dbl(x), x);
            return F::eq(lhs, F::add(F::norm(F::sub(x3, x3x)), m(be)));
        }
    }
};

}  // namespace ecgpu
