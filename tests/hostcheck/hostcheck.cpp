// hostcheck.cpp — TEST INFRASTRUCTURE.  Compiles the exact __host__ __device__ arithmetic the gfx950
// kernels use (csrc/ecgpu_field.h, ecgpu_point.h, ecgpu_recode.h) with g++ so that it can be checked
// against the oracle on a machine without a GPU.  The driver loops below mirror the kernels' control
// flow (k_fixed_base, k_var_base, k_normalize, the Pippenger pipeline) on one CPU thread.  Nothing here
// is linked into libecgpu.so.
#include <cstring>
#include <array>
#include <vector>

#include "../../elliptic-curves_amd/csrc/ecgpu_point.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_recode.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_fixedmul.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_varmul.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_ctmul.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_msm_chunk.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_scalar.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_sha256.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_hash.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_sm3.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_belt.h"
#include "../../elliptic-curves_amd/csrc/ecgpu_verify.h"

using namespace ecgpu;

namespace {

template <class C>
bool load_affine(Affine<C>* a, const uint8_t* xy, int inf) {
    using F = Field<C>;
    if (inf) return false;
    bool ok1, ok2;
    a->x = F::from_bytes(xy, &ok1).e;
    a->y = F::from_bytes(xy + WireBytes<C>::value, &ok2).e;
    return true;
}

template <class C>
void store_affine(const Proj<C>& p, uint8_t* xy, uint8_t* inf) {
    using F = Field<C>;
    using G = Group<C>;
    if (F::is_zero(G::m(p.z))) {
        std::memset(xy, 0, 2 * WireBytes<C>::value);
        if (inf) *inf = 1;
        return;
    }
    auto zi = F::inv(G::m(p.z));
    F::to_bytes(xy, F::mul(G::m(p.x), zi));
    F::to_bytes(xy + WireBytes<C>::value, F::mul(G::m(p.y), zi));
    if (inf) *inf = 0;
}

template <class C>
typename Field<C>::M1 times21(const typename Field<C>::M1& y) {
    using F = Field<C>;
    if constexpr (C::REPR == REPR_U29_K256) {
        return F::template mul_small<21>(y);
    } else {
        uint32_t w[C::N] = {21};
        return F::mul(y, F::from_canonical(w));
    }
}

template <class C>
int field_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    using F = Field<C>;
    bool ok;
    auto x = F::from_bytes(a, &ok);
    if (!ok) return -3;
    auto y = F::zero();
    if (b) { y = F::from_bytes(b, &ok); if (!ok) return -3; }
    switch (op) {
    case 0: F::to_bytes(out, F::add(x, y)); break;
    case 1: F::to_bytes(out, F::norm(F::sub(x, y))); break;
    case 2: F::to_bytes(out, F::mul(x, y)); break;
    case 3: F::to_bytes(out, F::sqr(x)); break;
    case 4: F::to_bytes(out, F::inv(x)); break;
    case 10: F::to_bytes(out, F::inv_fermat(x)); break;    // independent check of the safegcd inversion
    case 17: F::to_bytes(out, F::template inv<true>(x)); break;    // the variable-time division steps (ModInv::invert_var)
    case 11: {                                            // square root: the root, or zero if there is none
        bool root;
        auto r = F::sqrt(x, &root);
        F::to_bytes(out, root ? r : F::zero());
        break;
    }
    case 5: F::to_bytes(out, F::neg(x)); break;
    case 6: F::to_bytes(out, times21<C>(x)); break;
    case 7: F::to_bytes(out, F::dbl(x)); break;
    case 8: {   // pack / unpack round trip of a lazy value
        uint32_t w[C::N];
        F::pack(w, F::norm(F::add(F::dbl(x), y)));
        F::to_bytes(out, F::unpack(w));
        break;
    }
    case 9: F::to_bytes(out, F::mul2(x, y, F::add(x, y), F::neg(y))); break;   // x*y - (x+y)*y
    case 13:                                                                   // x y - (2 x + y): the fused F::mul_sub (k256)
        if constexpr (C::REPR == ecgpu::REPR_U29_K256) F::to_bytes(out, F::mul_sub(x, y, F::add(F::dbl(x), y)));
        else F::to_bytes(out, F::norm(F::sub(F::mul(x, y), F::add(F::dbl(x), y))));
        break;
    case 14:                                                                   // (x + y)^2 - 5 y: the fused F::sqr_sub (k256)
        if constexpr (C::REPR == ecgpu::REPR_U29_K256) F::to_bytes(out, F::sqr_sub(F::add(x, y), F::add(y, F::dbl(F::dbl(y)))));
        else F::to_bytes(out, F::norm(F::sub(F::sqr(F::norm(F::add(x, y))), F::add(y, F::dbl(F::dbl(y))))));
        break;
    default: return -1;
    }
    return 0;
}

// chains of operations keep values in their lazy internal form between steps and run the magnitudes
// up to the limits the point formulas use
template <class C>
int field_chain(const uint8_t* a, const uint8_t* b, int steps, uint8_t* out) {
    using F = Field<C>;
    bool ok;
    auto x = F::from_bytes(a, &ok), y = F::from_bytes(b, &ok);
    for (int i = 0; i < steps; i++) {
        auto t = F::mul(x, y);
        auto u = F::norm(F::sub(F::add(t, x), F::dbl(y)));
        auto v = F::norm(F::neg(F::add(t, times21<C>(y))));
        x = F::sqr(u);
        y = F::mul(v, F::one());          // back to value magnitude 1 for the Montgomery-lazy field
    }
    F::to_bytes(out, F::add(x, y));
    return 0;
}

template <class C>
int point_op(int op, const uint8_t* pxy, int pinf, const uint8_t* qxy, int qinf, uint8_t* out, uint8_t* oinf) {
    using G = Group<C>;
    auto b = G::curve_b();
    Affine<C> pa, qa;
    Proj<C> p = G::identity(), r;
    if (load_affine<C>(&pa, pxy, pinf)) p = G::from_affine(pa);
    bool qfinite = false;
    if (op == 0 || op == 1 || op == 4 || op == 5) qfinite = load_affine<C>(&qa, qxy, qinf);
    switch (op) {
    case 0: r = G::add(p, qfinite ? G::from_affine(qa) : G::identity(), b); break;
    case 1: r = qfinite ? G::add_mixed(p, qa, b) : p; break;
    case 2: r = G::dbl(p, b); break;
    case 3: r = G::neg(p); break;
    case 4: r = G::add(p, qfinite ? G::from_affine(qa) : G::identity(), b, true); break;     // p - q
    case 5: r = qfinite ? G::add_mixed(p, qa, b, true) : p; break;                            // p - q (mixed)
    default: return -1;
    }
    store_affine<C>(r, out, oinf);
    return 0;
}

template <class C>
int on_curve(const uint8_t* xy) {
    using G = Group<C>;
    using F = Field<C>;
    bool ok1, ok2;
    Affine<C> a;
    a.x = F::from_bytes(xy, &ok1).e;
    a.y = F::from_bytes(xy + WireBytes<C>::value, &ok2).e;
    return ok1 && ok2 && G::on_curve(a, G::curve_b());
}

// ---- mirrors of the kernels' control flow ----------------------------------------------------------

template <class C>
struct BaseTable {
    int w = 0, nwin = 0;
    std::vector<Affine<C>> e;   // [nwin][2^(w-1)]
};

// k_window_bases + k_table_entries + k_normalize<.., true>
template <class C>
void build_table(BaseTable<C>& t, int w) {
    using G = Group<C>;
    using F = Field<C>;
    auto b = G::curve_b();
    t.w = w;
    t.nwin = signed_window_count(32 * C::N - 1, w);
    size_t half = (size_t)1 << (w - 1);
    t.e.resize(half * t.nwin);
    Affine<C> g;
    g.x = F::from_canonical(C::GX).e;
    g.y = F::from_canonical(C::GY).e;
    Proj<C> base = G::from_affine(g);
    for (int j = 0; j < t.nwin; j++) {
        // running multiples instead of per-entry double-and-add: same group elements, cheaper on one core
        Proj<C> cur = base;
        for (size_t e = 0; e < half; e++) {
            auto zi = F::inv(G::m(cur.z));
            uint32_t w[C::N];                       // entries go through the packed storage form like on the GPU
            F::pack(w, F::mul(G::m(cur.x), zi));
            t.e[j * half + e].x = F::unpack(w).e;
            F::pack(w, F::mul(G::m(cur.y), zi));
            t.e[j * half + e].y = F::unpack(w).e;
            cur = G::add(cur, base, b);
        }
        for (int s = 0; s < w; s++) base = G::dbl(base, b);
    }
}

// k_table_entries' rule for entry e (1-based) of a window: lane t = (e - 1) mod T computes (t + 1) * base by
// double-and-add and then adds T * base once per stride
template <class C>
Proj<C> table_entry_rule(const Proj<C>& base, uint32_t e, int tlog = 3) {
    using G = Group<C>;
    auto b = G::curve_b();
    const uint32_t T = 1u << tlog, e0 = ((e - 1) & (T - 1)) + 1;
    Proj<C> step = base;
    for (int s = 0; s < tlog; s++) step = G::dbl(step, b);
    Proj<C> acc = base;
    for (int bit = 30 - __builtin_clz(e0); bit >= 0; bit--) {
        acc = G::dbl(acc, b);
        if ((e0 >> bit) & 1) acc = G::add(acc, base, b);
    }
    for (uint32_t x = e0; x < e; x += T) acc = G::add(acc, step, b);
    return acc;
}

// k_fixed_base: the shared per-lane body over the CPU-built table
template <class C>
struct BaseTableLocal {
    const BaseTable<C>* t;
    void load(PackedPoint<2 * C::N>& p, int window, uint32_t index) const {
        const Affine<C>& a = t->e[(size_t)window * ((size_t)1 << (t->w - 1)) + index];
        Field<C>::pack(p.w, Group<C>::m(a.x));
        Field<C>::pack(p.w + C::N, Group<C>::m(a.y));
    }
};
template <class C>
Proj<C> fixed_base_one(const BaseTable<C>& t, const uint32_t* k_in) {
    BaseTableLocal<C> tab{&t};
    return fixed_base_mul<C>(k_in, tab, t.w, t.nwin, Group<C>::curve_b());
}

// k_var_base: the shared per-lane body with a stack table
template <class C>
struct VarTabLocal {
    Fe<C::NL> t[8][3];
    void put_el(int e, int k, const Fe<C::NL>& v) { t[e][k] = v; }
    Fe<C::NL> get_el(int e, int k) const { return t[e][k]; }
};
template <class C>
Proj<C> var_base_one(const Affine<C>& a, const uint32_t* k) {
    VarTabLocal<C> tab;
    return var_base_mul<C>(a, k, Group<C>::curve_b(), tab);
}

// k_normalize<.., false> with `nthreads` strided lanes
template <class C>
void normalize(const std::vector<Proj<C>>& proj, size_t nthreads, uint8_t* out_xy, uint8_t* out_inf) {
    using F = Field<C>;
    using G = Group<C>;
    constexpr int N = C::N;
    size_t n = proj.size();
    std::vector<Fe<C::NL>> prefix(n);
    for (size_t t = 0; t < nthreads; t++) {
        typename F::M1 acc = F::one();
        for (size_t j = t; j < n; j += nthreads) {
            prefix[j] = acc.e;
            if (!F::is_zero(G::m(proj[j].z))) acc = F::mul(acc, G::m(proj[j].z));
        }
        typename F::M1 inv = F::inv(acc);
        if (n <= t) continue;
        size_t last = t + ((n - 1 - t) / nthreads) * nthreads;
        for (size_t j = last;; j -= nthreads) {
            const Proj<C>& p = proj[j];
            if (F::is_zero(G::m(p.z))) {
                std::memset(out_xy + j * 2 * WireBytes<C>::value, 0, 2 * WireBytes<C>::value);
                out_inf[j] = 1;
            } else {
                typename F::M1 zinv = F::mul(G::m(prefix[j]), inv);
                inv = F::mul(inv, G::m(p.z));
                F::to_bytes(out_xy + j * 2 * WireBytes<C>::value, F::mul(G::m(p.x), zinv));
                F::to_bytes(out_xy + j * 2 * WireBytes<C>::value + WireBytes<C>::value, F::mul(G::m(p.y), zinv));
                out_inf[j] = 0;
            }
            if (j < nthreads) break;
        }
    }
}

template <class C>
bool load_scalar(uint32_t* k, const uint8_t* be) {
    load_be_wire<C>(k, be);
    return !mp_geq<C::N>(k, C::ORDER);
}

template <class C>
int batch_mul_base(int w, const uint8_t* scalars, size_t n, size_t nthreads, uint8_t* out_xy, uint8_t* out_inf) {
    static BaseTable<C> table;
    if (table.w != w) build_table<C>(table, w);
    std::vector<Proj<C>> proj(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t k[C::N];
        if (!load_scalar<C>(k, scalars + i * WireBytes<C>::value)) return -2;
        proj[i] = fixed_base_one<C>(table, k);
    }
    normalize<C>(proj, nthreads ? nthreads : 1, out_xy, out_inf);
    return 0;
}

template <class C>
int batch_mul(const uint8_t* scalars, const uint8_t* pxy, const uint8_t* pinf, size_t n, size_t nthreads,
              uint8_t* out_xy, uint8_t* out_inf) {
    using G = Group<C>;
    std::vector<Proj<C>> proj(n);
    auto b = G::curve_b();
    for (size_t i = 0; i < n; i++) {
        uint32_t k[C::N];
        if (!load_scalar<C>(k, scalars + i * WireBytes<C>::value)) return -2;
        Affine<C> a;
        if (!load_affine<C>(&a, pxy + i * 2 * WireBytes<C>::value, pinf ? pinf[i] : 0)) { proj[i] = G::identity(); continue; }
        if (!G::on_curve(a, b)) return -3;
        proj[i] = var_base_one<C>(a, k);
    }
    normalize<C>(proj, nthreads ? nthreads : 1, out_xy, out_inf);
    return 0;
}

// k_fixed_base_ct / k_var_base_ct: the uniform-schedule bodies of ecgpu_ctmul.h (generator LUTs built the way
// ensure_ct_lut does: bases 2^(W i) G, 2^(W-1) multiples each, packed affine)
template <class C>
struct CtLutLocal {
    static constexpr bool UNIFORM = false;
    std::vector<uint32_t> w;     // [CT_BASE_LUTS][CT_BASE_ENTRIES][2 N]
    void load(PackedPoint<2 * C::N>& p, int i, int entry) const {
        std::memcpy(p.w, &w[((size_t)i * CT_BASE_ENTRIES + entry) * (2 * C::N)], 2 * C::N * 4);
    }
};
template <class C>
void build_ct_lut(CtLutLocal<C>& t) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N, NLUT = CT_BASE_LUTS<C>;
    auto b = G::curve_b();
    t.w.resize((size_t)NLUT * CT_BASE_ENTRIES * 2 * N);
    Affine<C> g;
    g.x = F::from_canonical(C::GX).e;
    g.y = F::from_canonical(C::GY).e;
    Proj<C> base = G::from_affine(g);
    for (int i = 0; i < NLUT; i++) {
        for (uint32_t e = 1; e <= (uint32_t)CT_BASE_ENTRIES; e++) {
            Proj<C> cur = table_entry_rule<C>(base, e, 0);
            auto zi = F::inv(G::m(cur.z));
            F::pack(&t.w[((size_t)i * CT_BASE_ENTRIES + e - 1) * 2 * N], F::mul(G::m(cur.x), zi));
            F::pack(&t.w[((size_t)i * CT_BASE_ENTRIES + e - 1) * 2 * N + N], F::mul(G::m(cur.y), zi));
        }
        for (int s = 0; s < CT_BASE_W; s++) base = G::dbl(base, b);
    }
}
template <class C>
int batch_mul_base_ct(const uint8_t* scalars, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    static CtLutLocal<C> lut;
    if (lut.w.empty()) build_ct_lut<C>(lut);
    std::vector<Proj<C>> proj(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t k[C::N];
        if (!load_scalar<C>(k, scalars + i * WireBytes<C>::value)) return -2;
        proj[i] = fixed_base_mul_ct<C>(k, lut, Group<C>::curve_b());
    }
    normalize<C>(proj, 1, out_xy, out_inf);
    return 0;
}
template <class C>
int batch_mul_ct(const uint8_t* scalars, const uint8_t* pxy, const uint8_t* pinf, size_t n, uint8_t* out_xy, uint8_t* out_inf) {
    using G = Group<C>;
    std::vector<Proj<C>> proj(n);
    auto b = G::curve_b();
    for (size_t i = 0; i < n; i++) {
        uint32_t k[C::N];
        if (!load_scalar<C>(k, scalars + i * WireBytes<C>::value)) return -2;
        Affine<C> a;
        Proj<C> p = G::identity();
        if (load_affine<C>(&a, pxy + i * 2 * WireBytes<C>::value, pinf ? pinf[i] : 0)) {
            if (!G::on_curve(a, b)) return -3;
            p = G::from_affine(a);
        }
        VarTabLocal<C> tab;
        proj[i] = var_base_mul_ct<C>(p, k, b, tab);
    }
    normalize<C>(proj, 1, out_xy, out_inf);
    return 0;
}

// the Pippenger pipeline of ecgpu_msm.h: prepare/scan/scatter/accumulate/reduce/combine, sequentially
template <class C, bool GLV>
int msm(int c, size_t chunk, const uint8_t* scalars, const uint8_t* pxy, const uint8_t* pinf, size_t n, uint8_t* out_xy,
        uint8_t* out_inf) {
    using G = Group<C>;
    using F = Field<C>;
    constexpr int N = C::N;
    auto b = G::curve_b();
    using S = MsmSplit<C, GLV>;
    const size_t npad = (n + 63) / 64 * 64, ne = (size_t)S::SUB * npad;      // sub-term h of term i sits at h * npad + i
    int nwin = signed_window_count(S::KBITS, c);
    size_t nb = (size_t)1 << (c - 1);
    int seg = nb < 32 ? (int)nb : 32;
    size_t nseg = nb / seg;
    std::vector<Affine<C>> pts(ne);
    std::vector<uint32_t> ranks((size_t)nwin * ne), sorted((size_t)nwin * ne + 16), counts((size_t)nwin * nb, 0),
        offsets((size_t)nwin * nb);
    std::vector<uint8_t> finite(n, 0);
    std::vector<std::array<std::array<uint32_t, S::KW>, S::SUB>> subs(n);
    std::vector<std::array<bool, S::SUB>> flips(n);
    auto digit_of = [&](size_t i, int h, int w, uint32_t* carry) {
        return msm_digit<S::KW>(subs[i][h].data(), w, c, nwin, carry, (uint32_t)(h * npad + i), flips[i][h], S::KBITS);
    };
    for (size_t i = 0; i < n; i++) {                                    // prepare
        uint32_t k[N];
        if (!load_scalar<C>(k, scalars + i * WireBytes<C>::value)) return -2;
        {
            uint32_t sub[S::SUB][S::KW];
            bool neg[S::SUB];
            S::split(k, sub, neg);
            for (int h = 0; h < S::SUB; h++) {
                for (int t = 0; t < S::KW; t++) subs[i][h][t] = sub[h][t];
                flips[i][h] = neg[h];
            }
        }
        if (!load_affine<C>(&pts[i], pxy + i * 2 * WireBytes<C>::value, pinf ? pinf[i] : 0)) continue;
        if (!G::on_curve(pts[i], b)) return -3;
        {
            uint32_t w[N];                                              // packed storage form, as on the GPU
            if constexpr (S::SUB == 2) {
                F::pack(w, F::mul(G::m(pts[i].x), F::unpack(C::BETA)));
                pts[npad + i].x = F::unpack(w).e;
            }
            F::pack(w, G::m(pts[i].x)); pts[i].x = F::unpack(w).e;
            F::pack(w, G::m(pts[i].y)); pts[i].y = F::unpack(w).e;
            if constexpr (S::SUB == 2) pts[npad + i].y = pts[i].y;
        }
        finite[i] = 1;
        for (int h = 0; h < S::SUB; h++) {
            uint32_t carry = 0;
            MsmDigitStream<S::KW> ds;                                   // what k_msm_prepare uses: must agree digit for digit
            ds.init(subs[i][h].data());
            for (int w = 0; w < nwin; w++) {
                MsmDigit d = digit_of(i, h, w, &carry);
                MsmDigit e = ds.next(w, c, nwin, (uint32_t)(h * npad + i), flips[i][h], S::KBITS);
                if (e.nonzero != d.nonzero || e.bucket != d.bucket || e.neg != d.neg) return -7;
                if (d.nonzero) ranks[(size_t)w * ne + h * npad + i] = counts[(size_t)w * nb + d.bucket]++;
            }
        }
    }
    for (int w = 0; w < nwin; w++) {                                    // scan
        uint32_t run = 0;
        for (size_t j = 0; j < nb; j++) { offsets[w * nb + j] = run; run += counts[w * nb + j]; }
    }
    for (size_t i = 0; i < n; i++) {                                    // scatter
        if (!finite[i]) continue;
        for (int h = 0; h < S::SUB; h++) {
            uint32_t carry = 0;
            for (int w = 0; w < nwin; w++) {
                MsmDigit d = digit_of(i, h, w, &carry);
                if (d.nonzero) {
                    const size_t j = h * npad + i;
                    uint32_t pos = offsets[(size_t)w * nb + d.bucket] + ranks[(size_t)w * ne + j];
                    sorted[(size_t)w * ne + pos] = (uint32_t)j | (d.neg << 31);
                }
            }
        }
    }
    // accumulate + finish: the lane bodies of k_msm_accumulate / k_msm_bucket_finish (ecgpu_msm_chunk.h)
    if (chunk == 0) chunk = 32;
    size_t nchunks = (ne + chunk - 1) / chunk;
    if (nchunks == 0) nchunks = 1;
    struct Points {
        const std::vector<Affine<C>>* pts;
        void load(PackedPoint<2 * N>& p, uint32_t term) const {
            F::pack(p.w, G::m((*pts)[term].x));
            F::pack(p.w + N, G::m((*pts)[term].y));
        }
    };
    struct Partials {
        std::vector<Xyzz<C>>* v;
        std::vector<uint8_t>* written;
        size_t base;
        void put(size_t slot, const Xyzz<C>& p) {
            if ((*written)[base + slot]) __builtin_trap();          // slots must be unique
            (*written)[base + slot] = 1;
            (*v)[base + slot] = p;
        }
        Xyzz<C> get(size_t slot) const {
            if (!(*written)[base + slot]) __builtin_trap();         // and every slot read must have been written
            return (*v)[base + slot];
        }
    };
    std::vector<Xyzz<C>> partials((size_t)nwin * (nb + nchunks));
    std::vector<uint8_t> written(partials.size(), 0);
    std::vector<Proj<C>> buckets((size_t)nwin * nb);
    for (int w = 0; w < nwin; w++) {
        const uint32_t* ow = offsets.data() + (size_t)w * nb;
        uint32_t total = ow[nb - 1] + counts[(size_t)w * nb + nb - 1];
        Points points{&pts};
        Partials sink{&partials, &written, (size_t)w * (nb + nchunks)};
        for (size_t q = 0; q < nchunks; q++)
            msm_chunk_accumulate<C>(sorted.data() + (size_t)w * ne, ow, total, (uint32_t)nb, (uint32_t)chunk, (uint32_t)q, b,
                                    points, sink);
        for (size_t j = 0; j < nb; j++)
            buckets[(size_t)w * nb + j] = msm_bucket_finish<C>((uint32_t)j, ow[j], counts[(size_t)w * nb + j], (uint32_t)chunk, b, sink,
                                                               sorted.data() + (size_t)w * ne, points);
    }
    std::vector<Proj<C>> wins(nwin);
    for (int w = 0; w < nwin; w++) {                                    // reduce
        Proj<C> wsum = G::identity();
        for (size_t s = 0; s < nseg; s++) {
            Proj<C> running = G::identity(), local = G::identity();
            size_t base = s * seg;
            const int sh = w == nwin - 1 ? msm_top_shift(S::KBITS, c) : 0;
            for (int j = seg - 1; j >= 0; j--) {
                running = G::add(running, buckets[(size_t)w * nb + base + j], b);
                if (j > 0 && ((base + j) >> sh) != ((base + j - 1) >> sh)) local = G::add(local, running, b);
            }
            {
                Proj<C> acc = G::identity();
                uint32_t k = (uint32_t)(base >> sh) + 1;
                int top = 31 - __builtin_clz(k);
                for (int bit = top; bit >= 0; bit--) {
                    acc = G::dbl(acc, b);
                    if ((k >> bit) & 1) acc = G::add(acc, running, b);
                }
                local = G::add(local, acc, b);
            }
            wsum = G::add(wsum, local, b);
        }
        wins[w] = wsum;
    }
    Proj<C> acc = wins[nwin - 1];                                       // combine
    for (int w = nwin - 2; w >= 0; w--) {
        for (int s = 0; s < c; s++) acc = G::dbl(acc, b);
        acc = G::add(acc, wins[w], b);
    }
    store_affine<C>(acc, out_xy, out_inf);
    return 0;
}

template <class C>
int msm_plain(int c, size_t chunk, const uint8_t* scalars, const uint8_t* pxy, const uint8_t* pinf, size_t n, uint8_t* out_xy,
              uint8_t* out_inf) {
    return msm<C, false>(c, chunk, scalars, pxy, pinf, n, out_xy, out_inf);
}

// ---- the verification / decompression kernels' per-element logic (ecgpu_verify.h) around the CPU mirrors of the
// fixed-base and variable-base kernels: k_ecdsa_prepare -> k_fixed_base + k_var_base (which adds its product to the fixed-base one) -> k_normalize ->
// k_ecdsa_finish, element by element
template <class C>
bool sum_affine_x(const BaseTable<C>& table, const uint32_t* a, const uint32_t* b, const uint32_t* cx, const uint32_t* cy,
                  uint32_t* x, uint32_t* y) {
    using F = Field<C>;
    using G = Group<C>;
    Affine<C> q;
    q.x = F::from_canonical(cx).e;
    q.y = F::from_canonical(cy).e;
    Proj<C> r = G::add(fixed_base_one<C>(table, a), var_base_one<C>(q, b), G::curve_b());
    if (F::is_zero(G::m(r.z))) return false;
    auto zi = F::inv(G::m(r.z));
    F::to_canonical(x, F::mul(G::m(r.x), zi));
    F::to_canonical(y, F::mul(G::m(r.y), zi));
    return true;
}

template <class C>
int ecdsa_verify(const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* q, size_t n, int reject_high_s, uint8_t* ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    static BaseTable<C> table;
    if (table.w != 8) build_table<C>(table, 8);
    for (size_t i = 0; i < n; i++) {
        uint32_t zw[N], rw[N], sw[N], cx[N], cy[N], u1[N], u2[N], x[N], y[N];
        load_be_wire<C>(zw, z + i * WB);
        load_be_wire<C>(rw, r + i * WB);
        load_be_wire<C>(sw, s + i * WB);
        load_be_wire<C>(cx, q + i * 2 * WB);
        load_be_wire<C>(cy, q + i * 2 * WB + WB);
        const bool valid = ecdsa_prepare_words<C>(zw, rw, sw, cx, cy, reject_high_s, u1, u2);
        const bool finite = sum_affine_x<C>(table, u1, u2, cx, cy, x, y);
        ok_out[i] = valid && finite && ecdsa_finish_words<C>(x, rw);
    }
    return 0;
}

// the per-element code of k_ecdsa_hash_msg: the curve's digest and bits2field (-> L wire bytes)
template <class C>
int ecdsa_hash_msg(const uint8_t* msgs, size_t msg_len, size_t n, uint8_t* z_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value, D = EcdsaDigest<C>::value;
    if constexpr (D == 0) {
        return -1;
    } else {
        for (size_t i = 0; i < n; i++) {
            uint8_t digest[D];
            const HashPiece one[1] = {{msgs + i * msg_len, msg_len}};
            sha2_pieces<D, 1>(digest, one);
            uint32_t zw[N];
            for (int j = 0; j < N; j++) zw[j] = 0;
            constexpr int TAKE = D < WB ? D : WB;
            for (int j = 0; j < TAKE; j++) {
                const int pos = TAKE - 1 - j;
                zw[pos / 4] |= (uint32_t)digest[j] << (8 * (pos % 4));
            }
            store_be_wire<C>(z_out + i * WB, zw);
        }
        return 0;
    }
}

// public-key recovery: the prepare logic of ecgpu_verify.h, the CPU mirrors of the two scalar multiplications, the finish rule
template <class C>
int ecdsa_recover(const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* recid, size_t n, int reject_high_s,
                  uint8_t* out_xy, uint8_t* ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    static BaseTable<C> table;
    if (table.w != 8) build_table<C>(table, 8);
    for (size_t i = 0; i < n; i++) {
        uint32_t zw[N], rw[N], sw[N], cx[N], cy[N], a[N], b[N], x[N], y[N];
        load_be_wire<C>(zw, z + i * WB);
        load_be_wire<C>(rw, r + i * WB);
        load_be_wire<C>(sw, s + i * WB);
        const bool valid = ecdsa_recover_prepare_words<C>(zw, rw, sw, recid[i], reject_high_s, a, b, cx, cy);
        const bool finite = sum_affine_x<C>(table, a, b, cx, cy, x, y);
        const bool ok = valid && finite;
        for (int j = 0; j < N; j++) { x[j] = ok ? x[j] : 0u; y[j] = ok ? y[j] : 0u; }
        store_be_wire<C>(out_xy + i * 2 * WB, x);
        store_be_wire<C>(out_xy + i * 2 * WB + WB, y);
        ok_out[i] = ok;
    }
    return 0;
}

// mode 0: challenge given (e, r, s, P); mode 1: from wire bytes (x-only key, message, 64-byte signature) — k256 only
int schnorr_verify(int mode, const uint8_t* e_or_msgs, size_t msg_len, const uint8_t* r_or_sigs, const uint8_t* s, const uint8_t* p,
                   size_t n, uint8_t* ok_out) {
    using C = K256Params;
    using S = ScalarN<C>;
    constexpr int N = 8;
    static BaseTable<C> table;
    if (table.w != 8) build_table<C>(table, 8);
    for (size_t i = 0; i < n; i++) {
        uint32_t ew[N], rw[N], sw[N], cx[N], cy[N], ne[N], x[N], y[N];
        bool valid;
        if (mode == 0) {
            load_be<N>(ew, e_or_msgs + 32 * i);
            load_be<N>(rw, r_or_sigs + 32 * i);
            load_be<N>(sw, s + 32 * i);
            load_be<N>(cx, p + 64 * i);
            load_be<N>(cy, p + 64 * i + 32);
            valid = schnorr_prepare_words<C>(ew, rw, sw, cx, cy, ne);
        } else {
            load_be<N>(cx, p + 32 * i);
            load_be<N>(rw, r_or_sigs + 64 * i);
            load_be<N>(sw, r_or_sigs + 64 * i + 32);
            valid = !mp_geq<N>(rw, C::P) && !S::is_zero(sw) && S::in_range(sw);
            valid = schnorr_lift_x<C>(cx, cy) && valid;
            Sha256::bip340_challenge(ew, r_or_sigs + 64 * i, p + 32 * i, e_or_msgs + i * msg_len, msg_len);
            schnorr_neg_challenge<C>(ne, ew);
            verify_blank<C>(valid, sw, ne, cx, cy);
        }
        const bool finite = sum_affine_x<C>(table, sw, ne, cx, cy, x, y);
        ok_out[i] = valid && finite && schnorr_finish_words<C>(x, y, rw);
    }
    return 0;
}

// k_sm2dsa_prepare -> s G + t Q -> k_sm2dsa_finish
int sm2dsa_verify(const uint8_t* e, const uint8_t* r, const uint8_t* s, const uint8_t* q, size_t n, uint8_t* ok_out) {
    using C = Sm2Params;
    constexpr int N = 8;
    static BaseTable<C> table;
    if (table.w != 8) build_table<C>(table, 8);
    for (size_t i = 0; i < n; i++) {
        uint32_t ew[N], rw[N], sw[N], cx[N], cy[N], t[N], x[N], y[N];
        load_be<N>(ew, e + 32 * i);
        load_be<N>(rw, r + 32 * i);
        load_be<N>(sw, s + 32 * i);
        load_be<N>(cx, q + 64 * i);
        load_be<N>(cy, q + 64 * i + 32);
        const bool valid = sm2dsa_prepare_words<C>(rw, sw, cx, cy, t);
        const bool finite = sum_affine_x<C>(table, sw, t, cx, cy, x, y);
        if (!finite) std::memset(x, 0, sizeof x);
        ok_out[i] = valid && sm2dsa_finish_words<C>(ew, x, !finite, rw);
    }
    return 0;
}

// `VerifyingKey::new(distid, pk)?.verify(msg, sig)`: the hashes of ecgpu_sm3.h, then the prehash logic above
int sm2dsa_verify_msg(const uint8_t* distid, size_t distid_len, const uint8_t* q, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs,
                      size_t n, uint8_t* ok_out) {
    using C = Sm2Params;
    for (size_t i = 0; i < n; i++) {
        uint32_t ew[8];
        Sm3::sm2_message_hash<C>(ew, distid, distid_len, q + 64 * i, msgs + i * msg_len, msg_len);
        uint8_t e[32];
        store_be<8>(e, ew);
        sm2dsa_verify(e, sigs + 64 * i, sigs + 64 * i + 32, q + 64 * i, 1, ok_out + i);
    }
    return 0;
}

// k_bign_prepare -> a G + b Q -> k_bign_finish (little-endian records: the words as they lie)
int bign_verify(const uint8_t* h, const uint8_t* sigs, const uint8_t* q, size_t n, uint8_t* ok_out) {
    using C = Bign256Params;
    constexpr int N = 8;
    static BaseTable<C> table;
    if (table.w != 8) build_table<C>(table, 8);
    auto le = [](uint32_t* w, const uint8_t* b, int nw) {
        for (int j = 0; j < nw; j++) w[j] = (uint32_t)b[4 * j] | (uint32_t)b[4 * j + 1] << 8 | (uint32_t)b[4 * j + 2] << 16 | (uint32_t)b[4 * j + 3] << 24;
    };
    for (size_t i = 0; i < n; i++) {
        uint32_t hw[N], s0w[4], s1w[N], cx[N], cy[N], a[N], b[N], x[N], y[N], t[8];
        le(hw, h + 32 * i, N);
        le(s0w, sigs + 48 * i, 4);
        le(s1w, sigs + 48 * i + 16, N);
        le(cx, q + 64 * i, N);
        le(cy, q + 64 * i + 32, N);
        const bool valid = bign_prepare_words<C>(hw, s0w, s1w, cx, cy, a, b);
        const bool finite = sum_affine_x<C>(table, a, b, cx, cy, x, y);
        uint8_t xb[32];
        for (int j = 0; j < 32; j++) xb[j] = (uint8_t)(x[j / 4] >> (8 * (j % 4)));
        const HashPiece pc[3] = {{Belt::OID, sizeof(Belt::OID)}, {xb, 32}, {h + 32 * i, 32}};
        Belt::hash_pieces<3>(Belt::H, t, pc);
        const bool eq = t[0] == s0w[0] && t[1] == s0w[1] && t[2] == s0w[2] && t[3] == s0w[3];
        ok_out[i] = valid && finite && eq;
    }
    return 0;
}
int bign_verify_msg(const uint8_t* q, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* ok_out) {
    for (size_t i = 0; i < n; i++) {
        uint32_t hw[8];
        const HashPiece pc[1] = {{msgs + i * msg_len, msg_len}};
        Belt::hash_pieces<1>(Belt::H, hw, pc);
        uint8_t h[32];
        for (int j = 0; j < 32; j++) h[j] = (uint8_t)(hw[j / 4] >> (8 * (j % 4)));
        bign_verify(h, sigs + 48 * i, q + 64 * i, 1, ok_out + i);
    }
    return 0;
}

template <class C>
int decompress(const uint8_t* xs, const uint8_t* odd, size_t n, uint8_t* out_xy, uint8_t* ok_out) {
    constexpr int N = C::N, WB = WireBytes<C>::value;
    for (size_t i = 0; i < n; i++) {
        uint32_t cx[N], cy[N];
        load_be_wire<C>(cx, xs + i * WB);
        ok_out[i] = decompress_words<C>(cx, odd[i] != 0, cy);
        store_be_wire<C>(out_xy + i * 2 * WB, cx);
        store_be_wire<C>(out_xy + i * 2 * WB + WB, cy);
    }
    return 0;
}

// ScalarN<C>: op 0 a*b mod n, 1 1/a mod n, 2 a mod n (a < 2^(32N)), 3 is_high(a) -> out[last byte]
template <class C>
int scalar_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    using S = ScalarN<C>;
    constexpr int N = C::N;
    uint32_t x[N], y[N], r[N];
    load_be_wire<C>(x, a);
    for (int i = 0; i < N; i++) r[i] = 0;
    if (b) load_be_wire<C>(y, b);
    switch (op) {
    case 0: S::mul(r, x, y); break;
    case 1: S::inv(r, x); break;
    case 2: S::reduce_wire(r, x); break;
    case 3: r[0] = S::is_high(x) ? 1u : 0u; break;
    default: return -1;
    }
    store_be_wire<C>(out, r);
    return 0;
}

template <class C>
int table_rule_check(int w, int j, uint32_t e, uint8_t* out_xy) {
    // e * 2^(w*j) * G via k_window_bases' doubling chain + k_table_entries' rule
    using G = Group<C>;
    using F = Field<C>;
    auto b = G::curve_b();
    Affine<C> g;
    g.x = F::from_canonical(C::GX).e;
    g.y = F::from_canonical(C::GY).e;
    Proj<C> base = G::from_affine(g);
    for (int s = 0; s < w * j; s++) base = G::dbl(base, b);
    uint8_t inf;
    store_affine<C>(table_entry_rule<C>(base, e), out_xy, &inf);
    return inf;
}

#define DISPATCH(curve, fn, args)                                                                                   \
    switch (curve) { case 0: return fn<K256Params> args; case 1: return fn<P256Params> args; case 2: return fn<P384Params> args; \
                     case 3: return fn<Sm2Params> args; case 4: return fn<P224Params> args; case 5: return fn<P192Params> args; case 6: return fn<P521Params> args; case 7: return fn<Bp256Params> args; case 8: return fn<Bp384Params> args; case 9: return fn<Bp256t1Params> args; case 10: return fn<Bp384t1Params> args; case 11: return fn<Bign256Params> args; default: return -1; }

}  // namespace

extern "C" {

int hc_field_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    DISPATCH(curve, field_op, (op, a, b, out))
}
// BIP340 challenge hash of r || pk || m -> 32 bytes big-endian
int hc_bip340_challenge(const uint8_t* r, const uint8_t* pk, const uint8_t* m, size_t msg_len, uint8_t* out) {
    uint32_t e[8];
    Sha256::bip340_challenge(e, r, pk, m, msg_len);
    store_be<8>(out, e);
    return 0;
}
int hc_scalar_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    DISPATCH(curve, scalar_op, (op, a, b, out))
}
int hc_field_chain(int curve, const uint8_t* a, const uint8_t* b, int steps, uint8_t* out) {
    DISPATCH(curve, field_chain, (a, b, steps, out))
}
int hc_point_op(int curve, int op, const uint8_t* p, int pi, const uint8_t* q, int qi, uint8_t* out, uint8_t* oi) {
    DISPATCH(curve, point_op, (op, p, pi, q, qi, out, oi))
}
int hc_on_curve(int curve, const uint8_t* xy) {
    DISPATCH(curve, on_curve, (xy))
}
int hc_batch_mul_base(int curve, int w, const uint8_t* s, size_t n, size_t nthreads, uint8_t* o, uint8_t* oi) {
    DISPATCH(curve, batch_mul_base, (w, s, n, nthreads, o, oi))
}
int hc_batch_mul(int curve, const uint8_t* s, const uint8_t* p, const uint8_t* pi, size_t n, size_t nthreads, uint8_t* o,
                 uint8_t* oi) {
    DISPATCH(curve, batch_mul, (s, p, pi, n, nthreads, o, oi))
}
int hc_batch_mul_base_ct(int curve, const uint8_t* s, size_t n, uint8_t* o, uint8_t* oi) {
    DISPATCH(curve, batch_mul_base_ct, (s, n, o, oi))
}
int hc_batch_mul_ct(int curve, const uint8_t* s, const uint8_t* p, const uint8_t* pi, size_t n, uint8_t* o, uint8_t* oi) {
    DISPATCH(curve, batch_mul_ct, (s, p, pi, n, o, oi))
}
// glv != 0: k256 on the GLV halves (MsmSplit<K256Params, true>), otherwise the plain folded scalar
int hc_msm(int curve, int c, size_t chunk, int glv, const uint8_t* s, const uint8_t* p, const uint8_t* pi, size_t n, uint8_t* o,
           uint8_t* oi) {
    if (glv) return curve == 0 ? msm<K256Params, true>(c, chunk, s, p, pi, n, o, oi) : -1;
    DISPATCH(curve, msm_plain, (c, chunk, s, p, pi, n, o, oi))
}
int hc_ecdsa_verify(int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* q, size_t n, int reject_high_s,
                    uint8_t* ok) {
    if (curve == 3 || curve == 11) return -1;                    // sm2 / bign signatures are not ECDSA
    DISPATCH(curve, ecdsa_verify, (z, r, s, q, n, reject_high_s, ok))
}
int hc_ecdsa_hash_msg(int curve, const uint8_t* msgs, size_t msg_len, size_t n, uint8_t* z_out) {
    DISPATCH(curve, ecdsa_hash_msg, (msgs, msg_len, n, z_out))
}
int hc_ecdsa_recover(int curve, const uint8_t* z, const uint8_t* r, const uint8_t* s, const uint8_t* recid, size_t n, int reject_high_s,
                     uint8_t* out_xy, uint8_t* ok) {
    if (curve == 3 || curve == 11) return -1;                    // sm2 / bign: not ECDSA
    DISPATCH(curve, ecdsa_recover, (z, r, s, recid, n, reject_high_s, out_xy, ok))
}
int hc_schnorr_verify(int mode, const uint8_t* e_or_msgs, size_t msg_len, const uint8_t* r_or_sigs, const uint8_t* s, const uint8_t* p,
                      size_t n, uint8_t* ok) {
    return schnorr_verify(mode, e_or_msgs, msg_len, r_or_sigs, s, p, n, ok);
}
int hc_sm2dsa_verify(const uint8_t* e, const uint8_t* r, const uint8_t* s, const uint8_t* q, size_t n, uint8_t* ok) {
    return sm2dsa_verify(e, r, s, q, n, ok);
}
int hc_sm2dsa_verify_msg(const uint8_t* distid, size_t distid_len, const uint8_t* q, const uint8_t* msgs, size_t msg_len,
                         const uint8_t* sigs, size_t n, uint8_t* ok) {
    return sm2dsa_verify_msg(distid, distid_len, q, msgs, msg_len, sigs, n, ok);
}
int hc_bign_verify(const uint8_t* h, const uint8_t* sigs, const uint8_t* q, size_t n, uint8_t* ok) { return bign_verify(h, sigs, q, n, ok); }
int hc_bign_verify_msg(const uint8_t* q, const uint8_t* msgs, size_t msg_len, const uint8_t* sigs, size_t n, uint8_t* ok) {
    return bign_verify_msg(q, msgs, msg_len, sigs, n, ok);
}
// belt-hash of an arbitrary message, optionally as two pieces split at `cut`, through the device-side absorber
int hc_belt_hash(const uint8_t* msg, size_t len, size_t cut, uint8_t* out32) {
    uint32_t d[8];
    if (cut > len) cut = len;
    const HashPiece pc[2] = {{msg, cut}, {msg + cut, len - cut}};
    Belt::hash_pieces<2>(Belt::H, d, pc);
    for (int j = 0; j < 32; j++) out32[j] = (uint8_t)(d[j / 4] >> (8 * (j % 4)));
    return 0;
}
// SM3 of an arbitrary message through the device-side absorber
int hc_sm3(const uint8_t* msg, size_t len, uint8_t* out32) {
    uint32_t d[8];
    Sm3::hash(d, msg, len);
    for (int i = 0; i < 8; i++) { out32[4 * i] = d[i] >> 24; out32[4 * i + 1] = d[i] >> 16; out32[4 * i + 2] = d[i] >> 8; out32[4 * i + 3] = d[i]; }
    return 0;
}
int hc_decompress(int curve, const uint8_t* xs, const uint8_t* odd, size_t n, uint8_t* out_xy, uint8_t* ok) {
    DISPATCH(curve, decompress, (xs, odd, n, out_xy, ok))
}
int hc_table_rule(int curve, int w, int j, uint32_t e, uint8_t* out_xy) {
    DISPATCH(curve, table_rule_check, (w, j, e, out_xy))
}
// Radix16Msb digits of a big-endian scalar (nl limbs), ndigits = 8*nl + 1, least significant first
int hc_radix16(const uint8_t* be, int nl, int8_t* digits) {
    if (nl == 8) {
        uint32_t k[8]; load_be<8>(k, be);
        Radix16Msb<8> r; r.init(k);
        for (int i = 0; i <= 64; i++) digits[i] = (int8_t)r.digit(i);
    } else if (nl == 12) {
        uint32_t k[12]; load_be<12>(k, be);
        Radix16Msb<12> r; r.init(k);
        for (int i = 0; i <= 96; i++) digits[i] = (int8_t)r.digit(i);
    } else return -1;
    return 0;
}
// signed w-bit window digits of a 256-bit big-endian scalar, returns nwin
int hc_signed_windows(const uint8_t* be, int w, int* digits) {
    uint32_t k[8]; load_be<8>(k, be);
    int nwin = signed_window_count(256, w);
    uint32_t carry = 0;
    for (int j = 0; j < nwin; j++) digits[j] = signed_window_step(get_bits<8>(k, j * w, w), w, &carry);
    return carry ? -1 : nwin;
}
int hc_k256_glv(const uint8_t* k_be, uint8_t* r1_be, uint8_t* r2_be) {
    uint32_t k[8], r1[8], r2[8];
    load_be<8>(k, k_be);
    K256Scalar::decompose(r1, r2, k);
    store_be<8>(r1_be, r1);
    store_be<8>(r2_be, r2);
    return (K256Scalar::is_high(r1) ? 1 : 0) | (K256Scalar::is_high(r2) ? 2 : 0);
}

}  // extern "C"
