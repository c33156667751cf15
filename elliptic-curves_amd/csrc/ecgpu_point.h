// ecgpu_point.h — short-Weierstrass group law on homogeneous projective coordinates with the
// Renes–Costello–Batina complete formulas, exactly the coordinate system and formulas the
// reference uses (so there are no exceptional cases to special-case on the GPU either):
//
//   a = 0  (k256)       k256/src/arithmetic/projective.rs:96-131 (Alg 7 add), :142-176 (Alg 8
//                       mixed add), :189-217 (Alg 9 double); b3 = 3*7 = 21 as a small constant
//   a = -3 (p256/p384)  primeorder/src/point_arithmetic.rs:222-245 (Alg 4), :254-280 (Alg 5),
//                       :289-318 (Alg 6)
//   identity            (0 : 1 : 0)   k256 projective.rs:49-53, primeorder projective.rs:60-64
//
// `__host__ __device__` for the same reason as ecgpu_field.h.
#pragma once

#include "ecgpu_field.h"

namespace ecgpu {

template <class C>
struct Affine {  // identity is NOT representable here; callers carry a flag / skip
    Fe<C::N> x, y;
};

template <class C>
struct Proj {
    Fe<C::N> x, y, z;
};

template <class C>
struct Group {
    using F = Field<C>;
    using E = Fe<C::N>;
    using P = Proj<C>;
    using A = Affine<C>;

    static ECGPU_HD P identity() {
        P r;
        r.x = F::zero();
        r.y = F::one();
        r.z = F::zero();
        return r;
    }
    static ECGPU_HD bool is_identity(const P& p) { return F::is_zero(p.z); }

    static ECGPU_HD P from_affine(const A& a) {
        P r;
        r.x = a.x;
        r.y = a.y;
        r.z = F::one();
        return r;
    }
    static ECGPU_HD P neg(const P& p) {
        P r = p;
        r.y = F::neg(p.y);
        return r;
    }
    static ECGPU_HD A neg(const A& p) {
        A r = p;
        r.y = F::neg(p.y);
        return r;
    }
    static ECGPU_HD E curve_b() {  // Montgomery form of b for the a = -3 curves
        E b;
        if constexpr (C::MONTGOMERY) {
#pragma unroll
            for (int i = 0; i < C::N; i++) b.v[i] = C::B[i];
            b = F::from_canonical(b);
        } else {
            b = F::zero();
            b.v[0] = C::B_SMALL;
        }
        return b;
    }

    // ---- a = 0 -------------------------------------------------------------------------------
    static ECGPU_HD P add_a0(const P& p, const P& q) {
        const uint32_t b3 = 3 * C::B_SMALL;
        E xx = F::mul(p.x, q.x);
        E yy = F::mul(p.y, q.y);
        E zz = F::mul(p.z, q.z);
        E xy = F::sub(F::mul(F::add(p.x, p.y), F::add(q.x, q.y)), F::add(xx, yy));
        E yz = F::sub(F::mul(F::add(p.y, p.z), F::add(q.y, q.z)), F::add(yy, zz));
        E xz = F::sub(F::mul(F::add(p.x, p.z), F::add(q.x, q.z)), F::add(xx, zz));
        E bzz3 = F::mul_small(zz, b3);
        E yy_m = F::sub(yy, bzz3);
        E yy_p = F::add(yy, bzz3);
        E byz3 = F::mul_small(yz, b3);
        E xx3 = F::add(F::dbl(xx), xx);
        E bxx9 = F::mul_small(xx3, b3);
        P r;
        r.x = F::sub(F::mul(xy, yy_m), F::mul(byz3, xz));
        r.y = F::add(F::mul(yy_p, yy_m), F::mul(bxx9, xz));
        r.z = F::add(F::mul(yz, yy_p), F::mul(xx3, xy));
        return r;
    }
    static ECGPU_HD P add_mixed_a0(const P& p, const A& q) {
        const uint32_t b3 = 3 * C::B_SMALL;
        E xx = F::mul(p.x, q.x);
        E yy = F::mul(p.y, q.y);
        E xy = F::sub(F::mul(F::add(p.x, p.y), F::add(q.x, q.y)), F::add(xx, yy));
        E yz = F::add(F::mul(q.y, p.z), p.y);
        E xz = F::add(F::mul(q.x, p.z), p.x);
        E bzz3 = F::mul_small(p.z, b3);
        E yy_m = F::sub(yy, bzz3);
        E yy_p = F::add(yy, bzz3);
        E byz3 = F::mul_small(yz, b3);
        E xx3 = F::add(F::dbl(xx), xx);
        E bxx9 = F::mul_small(xx3, b3);
        P r;
        r.x = F::sub(F::mul(xy, yy_m), F::mul(byz3, xz));
        r.y = F::add(F::mul(yy_p, yy_m), F::mul(bxx9, xz));
        r.z = F::add(F::mul(yz, yy_p), F::mul(xx3, xy));
        return r;
    }
    static ECGPU_HD P dbl_a0(const P& p) {
        const uint32_t b3 = 3 * C::B_SMALL;
        E yy = F::sqr(p.y);
        E zz = F::sqr(p.z);
        E xy2 = F::dbl(F::mul(p.x, p.y));
        E bzz3 = F::mul_small(zz, b3);
        E bzz9 = F::add(F::dbl(bzz3), bzz3);
        E yy_m9 = F::sub(yy, bzz9);
        E yy_p3 = F::add(yy, bzz3);
        E t = F::mul_small(F::mul(yy, zz), 8 * b3);  // 24*b*yy*zz
        P r;
        r.x = F::mul(xy2, yy_m9);
        E yyy_z = F::mul(F::mul(yy, p.y), p.z);
        r.z = F::dbl(F::dbl(F::dbl(yyy_z)));
        r.y = F::add(F::mul(yy_m9, yy_p3), t);
        return r;
    }

    // ---- a = -3 ------------------------------------------------------------------------------
    static ECGPU_HD P add_am3(const P& l, const P& r, const E& b) {
        E xx = F::mul(l.x, r.x);
        E yy = F::mul(l.y, r.y);
        E zz = F::mul(l.z, r.z);
        E xy = F::sub(F::mul(F::add(l.x, l.y), F::add(r.x, r.y)), F::add(xx, yy));
        E yz = F::sub(F::mul(F::add(l.y, l.z), F::add(r.y, r.z)), F::add(yy, zz));
        E xz = F::sub(F::mul(F::add(l.x, l.z), F::add(r.x, r.z)), F::add(xx, zz));
        E bzz = F::sub(xz, F::mul(b, zz));
        E bzz3 = F::add(F::dbl(bzz), bzz);
        E yy_m = F::sub(yy, bzz3);
        E yy_p = F::add(yy, bzz3);
        E zz3 = F::add(F::dbl(zz), zz);
        E bxz = F::sub(F::mul(b, xz), F::add(zz3, xx));
        E bxz3 = F::add(F::dbl(bxz), bxz);
        E xx3_m_zz3 = F::sub(F::add(F::dbl(xx), xx), zz3);
        P o;
        o.x = F::sub(F::mul(yy_p, xy), F::mul(yz, bxz3));
        o.y = F::add(F::mul(yy_p, yy_m), F::mul(xx3_m_zz3, bxz3));
        o.z = F::add(F::mul(yy_m, yz), F::mul(xy, xx3_m_zz3));
        return o;
    }
    static ECGPU_HD P add_mixed_am3(const P& l, const A& r, const E& b) {
        E xx = F::mul(l.x, r.x);
        E yy = F::mul(l.y, r.y);
        E xy = F::sub(F::mul(F::add(l.x, l.y), F::add(r.x, r.y)), F::add(xx, yy));
        E yz = F::add(F::mul(r.y, l.z), l.y);
        E xz = F::add(F::mul(r.x, l.z), l.x);
        E bz = F::sub(xz, F::mul(b, l.z));
        E bz3 = F::add(F::dbl(bz), bz);
        E yy_m = F::sub(yy, bz3);
        E yy_p = F::add(yy, bz3);
        E z3 = F::add(F::dbl(l.z), l.z);
        E bxz = F::sub(F::mul(b, xz), F::add(z3, xx));
        E bxz3 = F::add(F::dbl(bxz), bxz);
        E xx3_m_zz3 = F::sub(F::add(F::dbl(xx), xx), z3);
        P o;
        o.x = F::sub(F::mul(yy_p, xy), F::mul(yz, bxz3));
        o.y = F::add(F::mul(yy_p, yy_m), F::mul(xx3_m_zz3, bxz3));
        o.z = F::add(F::mul(yy_m, yz), F::mul(xy, xx3_m_zz3));
        return o;
    }
    static ECGPU_HD P dbl_am3(const P& p, const E& b) {
        E xx = F::sqr(p.x);
        E yy = F::sqr(p.y);
        E zz = F::sqr(p.z);
        E xy2 = F::dbl(F::mul(p.x, p.y));
        E xz2 = F::dbl(F::mul(p.x, p.z));
        E bzz = F::sub(F::mul(b, zz), xz2);
        E bzz3 = F::add(F::dbl(bzz), bzz);
        E yy_m = F::sub(yy, bzz3);
        E yy_p = F::add(yy, bzz3);
        E y_frag = F::mul(yy_p, yy_m);
        E x_frag = F::mul(yy_m, xy2);
        E zz3 = F::add(F::dbl(zz), zz);
        E bxz2 = F::sub(F::mul(b, xz2), F::add(zz3, xx));
        E bxz6 = F::add(F::dbl(bxz2), bxz2);
        E xx3_m_zz3 = F::sub(F::add(F::dbl(xx), xx), zz3);
        P o;
        o.y = F::add(y_frag, F::mul(xx3_m_zz3, bxz6));
        E yz2 = F::dbl(F::mul(p.y, p.z));
        o.x = F::sub(x_frag, F::mul(bxz6, yz2));
        o.z = F::dbl(F::dbl(F::mul(yz2, yy)));
        return o;
    }

    // ---- curve-generic entry points (b is ignored for a = 0) -----------------------------------
    static ECGPU_HD P add(const P& p, const P& q, const E& b) {
        if constexpr (C::A_IS_ZERO) return add_a0(p, q);
        else return add_am3(p, q, b);
    }
    static ECGPU_HD P add_mixed(const P& p, const A& q, const E& b) {
        if constexpr (C::A_IS_ZERO) return add_mixed_a0(p, q);
        else return add_mixed_am3(p, q, b);
    }
    static ECGPU_HD P dbl(const P& p, const E& b) {
        if constexpr (C::A_IS_ZERO) return dbl_a0(p);
        else return dbl_am3(p, b);
    }

    // y^2 == x^3 + a x + b   (primeorder/src/affine.rs:100-109)
    static ECGPU_HD bool on_curve(const A& p, const E& b) {
        E lhs = F::sqr(p.y);
        E x3 = F::mul(F::sqr(p.x), p.x);
        E rhs;
        if constexpr (C::A_IS_ZERO) {
            rhs = F::add(x3, b);
        } else {
            E x3x = F::add(F::dbl(p.x), p.x);
            rhs = F::add(F::sub(x3, x3x), b);
        }
        return F::eq(lhs, rhs);
    }
};

}  // namespace ecgpu
