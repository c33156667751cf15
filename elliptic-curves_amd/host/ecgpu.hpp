// ecgpu.hpp — host-side mirror (C++17, header-only) of the reference's scalar-multiplication surface,
// written above the C ABI of include/ecgpu.h.
//
// The reference is Rust; no Rust toolchain exists in this image, so the host side that a Rust user
// would see is mirrored here in C++ with the same names, argument meaning and error behaviour:
//
//   Scalar::from_repr                      k256/src/arithmetic/scalar.rs:310-316 (rejects >= n)
//   AffinePoint::{IDENTITY, from_coordinates, x, y}   primeorder/src/affine.rs:45-49,100-112
//   ProjectivePoint::{IDENTITY, to_affine, mul_by_generator, operator*, lincomb}
//                                          primeorder/src/projective.rs:60-86,133-137,480-511
//                                          k256/src/arithmetic/mul.rs:84-109,180-203,249-256
//   MulBackend<C>::{mul_by_generator, mul_by_generator_vartime, mul_by_generator_and_mul_add_vartime}
//                                          primeorder/src/mul_backend.rs:11-40
//   BatchNormalize::batch_normalize        primeorder/src/projective.rs:452-478
//   Sum for ProjectivePoint                primeorder/src/projective.rs (impl Sum)
//
// plus the batch forms that are the reason for a GPU backend (`batch_mul_by_generator`, `batch_mul`).
// A `ProjectivePoint` here stores the normalised (affine) coordinates: projective triples are
// algorithm-dependent and only canonical affine bytes cross the ABI.
// Single-element calls launch a batch of one on the GPU — correct, but the batch forms are the point.
// There is no CPU fallback: `Engine` throws `Error` when the device or the library is unusable.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ecgpu.h"

namespace ecgpu_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error("ecgpu error " + std::to_string(c) + ": " + what), code(c) {}
};

// One context per process/thread, the analogue of `static BASEPOINT_TABLE: LazyLock<..>`.
class Engine {
  public:
    explicit Engine(int device = 0) {
        int rc = ecgpu_init(&ctx_, device);
        if (rc != ECGPU_OK) throw Error(rc, "ecgpu_init failed (no gfx950 device?)");
    }
    ~Engine() { ecgpu_destroy(ctx_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    ecgpu_ctx* ctx() const { return ctx_; }
    void check(int rc) const {
        if (rc != ECGPU_OK) throw Error(rc, ecgpu_last_error(ctx_));
    }
    static Engine& global() {
        static thread_local Engine e(0);
        return e;
    }

  private:
    ecgpu_ctx* ctx_ = nullptr;
};

// Page-locked host bytes (ecgpu_host_alloc): batches kept here reach the GPU at PCIe DMA speed.
class PinnedBytes {
  public:
    PinnedBytes(Engine& e, size_t n) : e_(&e), n_(n), p_((uint8_t*)ecgpu_host_alloc(e.ctx(), n)) {
        if (!p_) throw Error(ECGPU_ERR_OOM, "ecgpu_host_alloc failed");
    }
    ~PinnedBytes() { ecgpu_host_free(e_->ctx(), p_); }
    PinnedBytes(const PinnedBytes&) = delete;
    PinnedBytes& operator=(const PinnedBytes&) = delete;
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return n_; }

  private:
    Engine* e_;
    size_t n_;
    uint8_t* p_;
};

template <int CURVE, size_t L>
struct Curve {
    static constexpr int ID = CURVE;
    static constexpr size_t FieldBytesSize = L;
    using FieldBytes = std::array<uint8_t, L>;

    // ---- Scalar ------------------------------------------------------------------------------------
    struct Scalar {
        FieldBytes repr{};   // canonical big-endian, < n (checked by the device on use)
        static Scalar from_repr(const FieldBytes& b) { return Scalar{b}; }
        static Scalar from_u64(uint64_t v) {
            Scalar s;
            for (int i = 0; i < 8; i++) s.repr[L - 1 - i] = (uint8_t)(v >> (8 * i));
            return s;
        }
        const FieldBytes& to_repr() const { return repr; }
    };

    // ---- AffinePoint -------------------------------------------------------------------------------
    struct AffinePoint {
        FieldBytes x_{}, y_{};
        uint8_t infinity = 1;
        static AffinePoint IDENTITY() { return AffinePoint{}; }
        static AffinePoint from_coordinates(const FieldBytes& x, const FieldBytes& y) {
            AffinePoint p;
            p.x_ = x; p.y_ = y; p.infinity = 0;
            return p;   // on-curve / range validation happens on the device (ECGPU_ERR_POINT)
        }
        const FieldBytes& x() const { return x_; }
        const FieldBytes& y() const { return y_; }
        bool is_identity() const { return infinity != 0; }
        bool operator==(const AffinePoint& o) const {
            return infinity == o.infinity && (infinity || (x_ == o.x_ && y_ == o.y_));
        }
        bool operator!=(const AffinePoint& o) const { return !(*this == o); }
    };

    // ---- ProjectivePoint ---------------------------------------------------------------------------
    struct ProjectivePoint {
        AffinePoint a;
        static ProjectivePoint IDENTITY() { return ProjectivePoint{}; }
        static ProjectivePoint from(const AffinePoint& p) { return ProjectivePoint{p}; }
        AffinePoint to_affine() const { return a; }
        bool is_identity() const { return a.is_identity(); }
        bool operator==(const ProjectivePoint& o) const { return a == o.a; }
        bool operator!=(const ProjectivePoint& o) const { return !(a == o.a); }

        // Group::mul_by_generator
        static ProjectivePoint mul_by_generator(const Scalar& k) { return batch_mul_by_generator({k})[0]; }
        static ProjectivePoint mul_by_generator_vartime(const Scalar& k) { return batch_mul_by_generator({k}, false)[0]; }
        // impl Mul<Scalar> / MulVartime
        ProjectivePoint operator*(const Scalar& k) const { return batch_mul({*this}, {k})[0]; }
        ProjectivePoint mul_vartime(const Scalar& k) const { return batch_mul({*this}, {k}, false)[0]; }
        // impl Add via Sum of two
        ProjectivePoint operator+(const ProjectivePoint& o) const { return sum({*this, o}); }

        // LinearCombination<[(ProjectivePoint, Scalar)]>::lincomb / lincomb_vartime
        // `lincomb` is the reference's constant-time name: the uniform-schedule entry point (one constant-time multiplication
        // per term + a tree of complete additions); `lincomb_vartime` is the bucket method, for public scalars
        static ProjectivePoint lincomb(const std::vector<std::pair<ProjectivePoint, Scalar>>& terms) {
            return lincomb_impl(terms, true);
        }
        static ProjectivePoint lincomb_vartime(const std::vector<std::pair<ProjectivePoint, Scalar>>& terms) {
            return lincomb_impl(terms, false);
        }
        static ProjectivePoint lincomb_impl(const std::vector<std::pair<ProjectivePoint, Scalar>>& terms, bool constant_time) {
            std::vector<uint8_t> s, p, f;
            pack(terms, s, p, f);
            AffinePoint out;
            Engine& e = Engine::global();
            uint8_t xy[2 * L];
            e.check((constant_time ? ecgpu_lincomb_ct : ecgpu_msm)(e.ctx(), ID, s.data(), p.data(), f.data(), terms.size(), xy,
                                                                   &out.infinity));
            std::memcpy(out.x_.data(), xy, L);
            std::memcpy(out.y_.data(), xy + L, L);
            return ProjectivePoint{out};
        }
        // MulByGeneratorVartime::mul_by_generator_and_mul_add_vartime: a*G + b*P
        static ProjectivePoint mul_by_generator_and_mul_add_vartime(const Scalar& a, const Scalar& b,
                                                                    const ProjectivePoint& p) {
            Engine& e = Engine::global();
            AffinePoint out;
            uint8_t pxy[2 * L], xy[2 * L];
            std::memcpy(pxy, p.a.x_.data(), L);
            std::memcpy(pxy + L, p.a.y_.data(), L);
            e.check(ecgpu_batch_mul_base_and_mul_add(e.ctx(), ID, a.repr.data(), b.repr.data(), pxy, &p.a.infinity, 1, xy,
                                                     &out.infinity));
            std::memcpy(out.x_.data(), xy, L);
            std::memcpy(out.y_.data(), xy + L, L);
            return ProjectivePoint{out};
        }
        // impl Sum for ProjectivePoint
        static ProjectivePoint sum(const std::vector<ProjectivePoint>& pts) {
            std::vector<uint8_t> p(pts.size() * 2 * L), f(pts.size());
            for (size_t i = 0; i < pts.size(); i++) {
                std::memcpy(&p[i * 2 * L], pts[i].a.x_.data(), L);
                std::memcpy(&p[i * 2 * L + L], pts[i].a.y_.data(), L);
                f[i] = pts[i].a.infinity;
                if (f[i]) std::memset(&p[i * 2 * L], 0, 2 * L);
            }
            Engine& e = Engine::global();
            AffinePoint out;
            uint8_t xy[2 * L];
            e.check(ecgpu_point_sum(e.ctx(), ID, p.data(), f.data(), pts.size(), xy, &out.infinity));
            std::memcpy(out.x_.data(), xy, L);
            std::memcpy(out.y_.data(), xy + L, L);
            return ProjectivePoint{out};
        }
    };

    static ProjectivePoint GENERATOR() { return ProjectivePoint::mul_by_generator(Scalar::from_u64(1)); }

    // ---- batch forms (new API; what the GPU is for) --------------------------------------------------
    // The names the reference uses for its constant-time operations (`mul_by_generator`, `Mul`, `diffie_hellman`) default to
    // the uniform-schedule entry points (ecgpu_batch_*_ct); the variable-time kernels (same results, 1.2-7x faster, for
    // PUBLIC scalars) stand behind the `*_vartime` names below, as in include/ecgpu.h and the Rust shim.
    static std::vector<ProjectivePoint> batch_mul_by_generator_vartime(const std::vector<Scalar>& ks) { return batch_mul_by_generator(ks, false); }
    static std::vector<ProjectivePoint> batch_mul_vartime(const std::vector<ProjectivePoint>& ps, const std::vector<Scalar>& ks) {
        return batch_mul(ps, ks, false);
    }
    static std::vector<FieldBytes> batch_diffie_hellman_vartime(const std::vector<Scalar>& secrets, const std::vector<AffinePoint>& publics) {
        return batch_diffie_hellman(secrets, publics, false);
    }
    static std::vector<ProjectivePoint> batch_mul_by_generator(const std::vector<Scalar>& ks, bool constant_time = true) {
        size_t n = ks.size();
        std::vector<uint8_t> s(n * L), xy(n * 2 * L), inf(n);
        for (size_t i = 0; i < n; i++) std::memcpy(&s[i * L], ks[i].repr.data(), L);
        Engine& e = Engine::global();
        e.check((constant_time ? ecgpu_batch_mul_base_ct : ecgpu_batch_mul_base)(e.ctx(), ID, s.data(), n, xy.data(), inf.data()));
        return unpack(xy, inf);
    }
    // k_i * G as SEC1 compressed points (tag || x, tag = 02 / 03, a single 00 byte's worth of zeros for the identity):
    // `(ProjectivePoint::mul_by_generator(k)).to_affine().to_sec1_point(true)` for a batch
    using CompressedPoint = std::array<uint8_t, L + 1>;
    static std::vector<CompressedPoint> batch_mul_by_generator_compressed(const std::vector<Scalar>& ks) {
        size_t n = ks.size();
        std::vector<uint8_t> s(n * L), x(n * L), tag(n);
        for (size_t i = 0; i < n; i++) std::memcpy(&s[i * L], ks[i].repr.data(), L);
        Engine& e = Engine::global();
        e.check(ecgpu_batch_mul_base_compressed(e.ctx(), ID, s.data(), n, x.data(), tag.data()));
        std::vector<CompressedPoint> out(n);
        for (size_t i = 0; i < n; i++) {
            out[i][0] = tag[i];
            std::memcpy(&out[i][1], &x[i * L], L);
        }
        return out;
    }
    static std::vector<ProjectivePoint> batch_mul(const std::vector<ProjectivePoint>& ps, const std::vector<Scalar>& ks,
                                                  bool constant_time = true) {
        size_t n = ks.size();
        if (ps.size() != n) throw Error(ECGPU_ERR_ARG, "batch_mul: length mismatch");
        std::vector<uint8_t> s(n * L), p(n * 2 * L), f(n), xy(n * 2 * L), inf(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&s[i * L], ks[i].repr.data(), L);
            std::memcpy(&p[i * 2 * L], ps[i].a.x_.data(), L);
            std::memcpy(&p[i * 2 * L + L], ps[i].a.y_.data(), L);
            f[i] = ps[i].a.infinity;
            if (f[i]) std::memset(&p[i * 2 * L], 0, 2 * L);
        }
        Engine& e = Engine::global();
        e.check((constant_time ? ecgpu_batch_mul_ct : ecgpu_batch_mul)(e.ctx(), ID, s.data(), p.data(), f.data(), n, xy.data(), inf.data()));
        return unpack(xy, inf);
    }

    // ---- the callers either side of the path, batch forms ------------------------------
    // DecompressPoint::decompress(x_bytes, y_is_odd) -> CtOption<AffinePoint>   (primeorder/src/affine.rs:183-200)
    static std::vector<AffinePoint> batch_decompress(const std::vector<FieldBytes>& xs, const std::vector<uint8_t>& y_is_odd) {
        size_t n = xs.size();
        if (y_is_odd.size() != n) throw Error(ECGPU_ERR_ARG, "batch_decompress: length mismatch");
        std::vector<uint8_t> x(n * L), xy(n * 2 * L), ok(n);
        for (size_t i = 0; i < n; i++) std::memcpy(&x[i * L], xs[i].data(), L);
        Engine& e = Engine::global();
        e.check(ecgpu_batch_decompress(e.ctx(), ID, x.data(), y_is_odd.data(), n, xy.data(), ok.data()));
        std::vector<AffinePoint> out(n);                     // None is reported as the identity flag
        for (size_t i = 0; i < n; i++) {
            if (!ok[i]) continue;
            std::memcpy(out[i].x_.data(), &xy[i * 2 * L], L);
            std::memcpy(out[i].y_.data(), &xy[i * 2 * L + L], L);
            out[i].infinity = 0;
        }
        return out;
    }
    // elliptic_curve::ecdh::diffie_hellman(secret, public).raw_secret_bytes()   ({k256,p256,p384}/src/ecdh.rs)
    static std::vector<FieldBytes> batch_diffie_hellman(const std::vector<Scalar>& secrets, const std::vector<AffinePoint>& publics,
                                                        bool constant_time = true) {
        size_t n = secrets.size();
        if (publics.size() != n) throw Error(ECGPU_ERR_ARG, "batch_diffie_hellman: length mismatch");
        std::vector<uint8_t> s(n * L), p(n * 2 * L), x(n * L), ok(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&s[i * L], secrets[i].repr.data(), L);
            std::memcpy(&p[i * 2 * L], publics[i].x_.data(), L);
            std::memcpy(&p[i * 2 * L + L], publics[i].y_.data(), L);
        }
        Engine& e = Engine::global();
        e.check((constant_time ? ecgpu_batch_ecdh_ct : ecgpu_batch_ecdh)(e.ctx(), ID, s.data(), p.data(), n, x.data(), ok.data()));
        std::vector<FieldBytes> out(n);
        for (size_t i = 0; i < n; i++) std::memcpy(out[i].data(), &x[i * L], L);
        return out;
    }
    // ecdsa::hazmat::verify_prehashed for a batch: (z, r, s) as field-sized big-endian integers, q the public keys;
    // normalize_s = the curve's EcdsaCurve::NORMALIZE_S (k256/src/ecdsa.rs:104-106)
    struct EcdsaSignature {
        FieldBytes r{}, s{};
    };
    static std::vector<uint8_t> batch_verify_prehashed(const std::vector<AffinePoint>& q, const std::vector<FieldBytes>& z,
                                                       const std::vector<EcdsaSignature>& sig, bool normalize_s) {
        size_t n = q.size();
        if (z.size() != n || sig.size() != n) throw Error(ECGPU_ERR_ARG, "batch_verify_prehashed: length mismatch");
        std::vector<uint8_t> zb(n * L), rb(n * L), sb(n * L), qb(n * 2 * L), ok(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&zb[i * L], z[i].data(), L);
            std::memcpy(&rb[i * L], sig[i].r.data(), L);
            std::memcpy(&sb[i * L], sig[i].s.data(), L);
            std::memcpy(&qb[i * 2 * L], q[i].x_.data(), L);
            std::memcpy(&qb[i * 2 * L + L], q[i].y_.data(), L);
        }
        Engine& e = Engine::global();
        e.check(ecgpu_ecdsa_verify_batch(e.ctx(), ID, zb.data(), rb.data(), sb.data(), qb.data(), n, normalize_s ? 1 : 0, ok.data()));
        return ok;
    }

    // ecdsa `VerifyingKey::recover_from_prehash` for a batch (reference vectors: k256/src/ecdsa.rs:190-262): the key each
    // signature recovers to under its `RecoveryId::to_byte`, or the identity (`is_identity()`) where recovery fails
    static std::vector<AffinePoint> batch_recover_from_prehash(const std::vector<FieldBytes>& z, const std::vector<EcdsaSignature>& sig,
                                                               const std::vector<uint8_t>& recovery_id, bool normalize_s) {
        size_t n = z.size();
        if (sig.size() != n || recovery_id.size() != n) throw Error(ECGPU_ERR_ARG, "batch_recover_from_prehash: length mismatch");
        std::vector<uint8_t> zb(n * L), rb(n * L), sb(n * L), xy(n * 2 * L), ok(n);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&zb[i * L], z[i].data(), L);
            std::memcpy(&rb[i * L], sig[i].r.data(), L);
            std::memcpy(&sb[i * L], sig[i].s.data(), L);
        }
        Engine& e = Engine::global();
        e.check(ecgpu_ecdsa_recover_batch(e.ctx(), ID, zb.data(), rb.data(), sb.data(), recovery_id.data(), n, normalize_s ? 1 : 0,
                                          xy.data(), ok.data()));
        std::vector<AffinePoint> out(n);
        for (size_t i = 0; i < n; i++) {
            if (!ok[i]) continue;                               // stays AffinePoint::IDENTITY()
            std::memcpy(out[i].x_.data(), &xy[i * 2 * L], L);
            std::memcpy(out[i].y_.data(), &xy[i * 2 * L + L], L);
            out[i].infinity = 0;
        }
        return out;
    }

    // ---- MulBackend<C> plug-in (primeorder/src/mul_backend.rs:11-40) ----------------------------------
    struct GpuBackend {
        static ProjectivePoint mul_by_generator(const Scalar& k) { return ProjectivePoint::mul_by_generator(k); }
        static ProjectivePoint mul_by_generator_vartime(const Scalar& k) { return ProjectivePoint::mul_by_generator(k); }
        static ProjectivePoint mul_by_generator_and_mul_add_vartime(const Scalar& a, const Scalar& b,
                                                                    const ProjectivePoint& p) {
            return ProjectivePoint::mul_by_generator_and_mul_add_vartime(a, b, p);
        }
    };

  private:
    static void pack(const std::vector<std::pair<ProjectivePoint, Scalar>>& terms, std::vector<uint8_t>& s,
                     std::vector<uint8_t>& p, std::vector<uint8_t>& f) {
        size_t n = terms.size();
        s.assign(n * L + 1, 0); p.assign(n * 2 * L + 1, 0); f.assign(n + 1, 0);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&s[i * L], terms[i].second.repr.data(), L);
            f[i] = terms[i].first.a.infinity;
            if (!f[i]) {
                std::memcpy(&p[i * 2 * L], terms[i].first.a.x_.data(), L);
                std::memcpy(&p[i * 2 * L + L], terms[i].first.a.y_.data(), L);
            }
        }
    }
    static std::vector<ProjectivePoint> unpack(const std::vector<uint8_t>& xy, const std::vector<uint8_t>& inf) {
        std::vector<ProjectivePoint> out(inf.size());
        for (size_t i = 0; i < inf.size(); i++) {
            out[i].a.infinity = inf[i];
            if (!inf[i]) {
                std::memcpy(out[i].a.x_.data(), &xy[i * 2 * L], L);
                std::memcpy(out[i].a.y_.data(), &xy[i * 2 * L + L], L);
            }
        }
        return out;
    }
};

using k256 = Curve<ECGPU_K256, 32>;
using p256 = Curve<ECGPU_P256, 32>;
using p384 = Curve<ECGPU_P384, 48>;
using sm2 = Curve<ECGPU_SM2, 32>;
using p224 = Curve<ECGPU_P224, 28>;   // no point decompression (p = 1 mod 4)
using p192 = Curve<ECGPU_P192, 24>;
using p521 = Curve<ECGPU_P521, 66>;
using bp256r1 = Curve<ECGPU_BP256, 32>;
using bp384r1 = Curve<ECGPU_BP384, 48>;
using bp256t1 = Curve<ECGPU_BP256T1, 32>;
using bp384t1 = Curve<ECGPU_BP384T1, 48>;

}  // namespace ecgpu_host
