// test_projective.cpp — the reference's projective-point property tests, re-stated against the C++ host
// mirror (elliptic-curves_amd/host/ecgpu.hpp) so that they read like the originals:
//   k256/tests/projective.rs:75-140, p256/tests/projective.rs:83-148, p256/src/arithmetic/tables.rs:64-80
// Every operation below runs on the GPU through the C ABI.  Needs a gfx950 device (run by the -m gpu tests).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../elliptic-curves_amd/host/ecgpu.hpp"

using namespace ecgpu_host;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64() {  // SplitMix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static const uint8_t N_K256[32] = {0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xfe,
                                   0xba,0xae,0xdc,0xe6,0xaf,0x48,0xa0,0x3b,0xbf,0xd2,0x5e,0x8c,0xd0,0x36,0x41,0x41};
static const uint8_t N_P256[32] = {0xff,0xff,0xff,0xff,0x00,0x00,0x00,0x00,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,
                                   0xbc,0xe6,0xfa,0xad,0xa7,0x17,0x9e,0x84,0xf3,0xb9,0xca,0xc2,0xfc,0x63,0x25,0x51};
static const uint8_t N_P384[48] = {0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,
                                   0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xff,0xc7,0x63,0x4d,0x81,0xf4,0x37,0x2d,0xdf,
                                   0x58,0x1a,0x0d,0xb2,0x48,0xb0,0xa7,0x7a,0xec,0xec,0x19,0x6a,0xcc,0xc5,0x29,0x73};

// Scalar::reduce(bytes): one conditional subtraction of n (k256/tests/projective.rs:27-31)
template <class C>
typename C::Scalar random_scalar(const uint8_t* n_be) {
    constexpr size_t L = C::FieldBytesSize;
    typename C::FieldBytes b;
    for (size_t i = 0; i < L; i += 8) {
        uint64_t v = next_u64();
        std::memcpy(&b[i], &v, 8);
    }
    if (std::memcmp(b.data(), n_be, L) >= 0) {
        int borrow = 0;
        for (int i = (int)L - 1; i >= 0; i--) {
            int d = (int)b[i] - (int)n_be[i] - borrow;
            borrow = d < 0;
            b[i] = (uint8_t)(d + (borrow << 8));
        }
    }
    return C::Scalar::from_repr(b);
}

// ECDSA vectors of the reference (tests/golden/<curve>.json), flattened by tests/test_gpu_cpp_mirror.py into
// "<qx> <qy> <z> <r> <s>" hex lines in the file named by $ECGPU_ECDSA_VECTORS_<curve id>
template <class C>
struct GoldenEcdsa {
    std::vector<typename C::AffinePoint> q;
    std::vector<typename C::FieldBytes> z;
    std::vector<typename C::EcdsaSignature> sig;
};
template <class C>
GoldenEcdsa<C> load_golden_ecdsa() {
    GoldenEcdsa<C> g;
    char name[64];
    std::snprintf(name, sizeof name, "ECGPU_ECDSA_VECTORS_%d", C::ID);
    const char* path = std::getenv(name);
    if (!path) return g;
    std::FILE* f = std::fopen(path, "r");
    if (!f) return g;
    constexpr size_t L = C::FieldBytesSize;
    char buf[5][2 * 48 + 2];
    while (std::fscanf(f, "%98s %98s %98s %98s %98s", buf[0], buf[1], buf[2], buf[3], buf[4]) == 5) {
        typename C::FieldBytes v[5];
        for (int k = 0; k < 5; k++)
            for (size_t i = 0; i < L; i++) {
                unsigned byte = 0;
                std::sscanf(buf[k] + 2 * i, "%2x", &byte);
                v[k][i] = (uint8_t)byte;
            }
        g.q.push_back(C::AffinePoint::from_coordinates(v[0], v[1]));
        g.z.push_back(v[2]);
        typename C::EcdsaSignature sg;
        sg.r = v[3]; sg.s = v[4];
        g.sig.push_back(sg);
    }
    std::fclose(f);
    return g;
}

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

template <class C>
int run(const char* name, const uint8_t* n_be, int cases) {
    using Scalar = typename C::Scalar;
    using P = typename C::ProjectivePoint;
    const P G = C::GENERATOR();
    CHECK(!G.is_identity());

    for (int it = 0; it < cases; it++) {
        // fn projective() -> GENERATOR * scalar   (k256/tests/projective.rs:21-25)
        Scalar s1 = random_scalar<C>(n_be), s2 = random_scalar<C>(n_be), s3 = random_scalar<C>(n_be);
        auto pts = C::batch_mul_by_generator({random_scalar<C>(n_be), random_scalar<C>(n_be), random_scalar<C>(n_be)});
        P p1 = pts[0], p2 = pts[1], p3 = pts[2];

        // lincomb == p1*s1 + p2*s2 + p3*s3, and lincomb_vartime agrees
        P reference = (p1 * s1) + (p2 * s2) + (p3 * s3);
        P test = P::lincomb({{p1, s1}, {p2, s2}, {p3, s3}});
        CHECK(reference == test);
        CHECK(P::lincomb_vartime({{p1, s1}, {p2, s2}, {p3, s3}}) == reference);

        // mul_by_generator == GENERATOR * s, and the vartime variant
        CHECK(P::mul_by_generator(s1) == G * s1);
        CHECK(P::mul_by_generator_vartime(s2) == G * s2);

        // mul_vartime == p * s
        CHECK(p1.mul_vartime(s3) == p1 * s3);

        // mul_by_generator_and_mul_add_vartime(a, b, P) == G*a + P*b
        CHECK(P::mul_by_generator_and_mul_add_vartime(s1, s2, p3) == (G * s1) + (p3 * s2));
        CHECK(C::GpuBackend::mul_by_generator(s3) == G * s3);
    }

    // identities and edge scalars
    Scalar zero = Scalar::from_u64(0), one = Scalar::from_u64(1), two = Scalar::from_u64(2);
    CHECK((G * zero).is_identity());
    CHECK(G * one == G);
    CHECK(G * two == G + G);
    CHECK((P::IDENTITY() * two).is_identity());
    CHECK(P::mul_by_generator(zero).is_identity());
    CHECK(P::lincomb({}).is_identity());
    CHECK(P::lincomb({{P::IDENTITY(), two}, {G, one}}) == G);
    CHECK((G + P::IDENTITY()) == G);

    // decoding errors: scalar >= n is refused (Scalar::from_repr -> None in the reference)
    typename C::FieldBytes nb;
    std::memcpy(nb.data(), n_be, C::FieldBytesSize);
    bool threw = false;
    try {
        (void)P::mul_by_generator(Scalar::from_repr(nb));
    } catch (const Error& e) {
        threw = e.code == ECGPU_ERR_SCALAR_RANGE;
    }
    CHECK(threw);
    // a point that is not on the curve is refused (AffinePoint::from_coordinates -> None)
    auto bad = G.to_affine();
    bad.y_[C::FieldBytesSize - 1] ^= 1;
    threw = false;
    try {
        (void)(P::from(bad) * two);
    } catch (const Error& e) {
        threw = e.code == ECGPU_ERR_POINT;
    }
    CHECK(threw);
    // decompress(compress(P)) == P, and ECDH is symmetric: x(a * (b G)) == x(b * (a G))
    {
        Scalar a = random_scalar<C>(n_be), b = random_scalar<C>(n_be);
        auto pub = C::batch_mul_by_generator({a, b});
        auto A = pub[0].to_affine(), B = pub[1].to_affine();
        auto back = C::batch_decompress({A.x(), B.x()}, {(uint8_t)(A.y()[C::FieldBytesSize - 1] & 1), (uint8_t)(B.y()[C::FieldBytesSize - 1] & 1)});
        CHECK(back[0] == A && back[1] == B);
        auto flipped = C::batch_decompress({A.x()}, {(uint8_t)((A.y()[C::FieldBytesSize - 1] & 1) ^ 1)});
        CHECK(flipped[0] != A && flipped[0].x() == A.x());
        auto shared = C::batch_diffie_hellman({a, b}, {B, A});
        CHECK(shared[0] == shared[1]);
        // the reference's names are the constant-time forms (uniform-schedule kernels): `Mul`, `mul_by_generator`,
        // `diffie_hellman` proper (primeorder/src/projective.rs:847-886, k256/src/arithmetic/mul.rs:180-197, k256/src/ecdh.rs:56-60);
        // the `*_vartime` names compute the same group elements on the variable-time kernels
        auto pub_vt = C::batch_mul_by_generator_vartime({a, b});
        CHECK(pub_vt[0] == pub[0] && pub_vt[1] == pub[1]);
        auto shared_vt = C::batch_diffie_hellman_vartime({a, b}, {B, A});
        CHECK(shared_vt[0] == shared[0] && shared_vt[1] == shared[1]);
        auto prod = C::batch_mul({pub[0], pub[1], P::IDENTITY()}, {b, a, a}), prod_vt = C::batch_mul_vartime({pub[0], pub[1], P::IDENTITY()}, {b, a, a});
        CHECK(prod_vt[0] == prod[0] && prod_vt[1] == prod[1] && prod_vt[0] == prod_vt[1] && prod_vt[2].is_identity());
        CHECK(P::mul_by_generator_vartime(a) == pub[0] && pub[1].mul_vartime(a) == prod[1]);
    }
    // ECDSA: a signature assembled from the verification equation itself verifies, a disturbed one does not.
    // Choose u1, u2, set R = u1 G + u2 Q, r = x(R) (when x(R) < n), s = r / u2, z = u1 s: then z/s = u1, r/s = u2.
    {
        Scalar d = random_scalar<C>(n_be);
        auto Q = P::mul_by_generator(d).to_affine();
        auto vec = load_golden_ecdsa<C>();
        if (!vec.q.empty()) {
            auto ok = C::batch_verify_prehashed(vec.q, vec.z, vec.sig, false);
            for (auto v : ok) CHECK(v == 1);
            auto bad = vec.sig;
            for (auto& sg : bad) sg.s[C::FieldBytesSize - 1] ^= 1;
            ok = C::batch_verify_prehashed(vec.q, vec.z, bad, false);
            for (auto v : ok) CHECK(v == 0);
            // recover_from_prehash: one of the two parities of the unreduced candidate gives back the signer's key
            // (the vectors carry no recovery id); the recovered keys are finite and verify
            std::vector<uint8_t> id0(vec.q.size(), 0), id1(vec.q.size(), 1);
            auto k0 = C::batch_recover_from_prehash(vec.z, vec.sig, id0, false);
            auto k1 = C::batch_recover_from_prehash(vec.z, vec.sig, id1, false);
            for (size_t i = 0; i < vec.q.size(); i++) CHECK((k0[i] == vec.q[i]) != (k1[i] == vec.q[i]));
            std::vector<uint8_t> bad_id(vec.q.size(), 7);
            for (auto& k : C::batch_recover_from_prehash(vec.z, vec.sig, bad_id, false)) CHECK(k.is_identity());
        }
        (void)Q;
    }
    std::printf("%s: %d proptest cases + edge cases ok\n", name, cases);
    return 0;
}

int main(int argc, char** argv) {
    int cases = argc > 1 ? std::atoi(argv[1]) : 8;
    try {
        if (run<k256>("k256", N_K256, cases)) return 1;
        if (run<p256>("p256", N_P256, cases)) return 1;
        if (run<p384>("p384", N_P384, cases)) return 1;
    } catch (const Error& e) {
        std::fprintf(stderr, "unexpected %s\n", e.what());
        return 2;
    }
    std::printf("all ok\n");
    return 0;
}
