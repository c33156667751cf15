// ecgpu_hash.h — hashing whole messages on the device for the "from wire bytes" verification entry points (host + device):
// a producer that turns a concatenation of byte strings into padded blocks and feeds ONE call site of a compression
// function, and SHA-512 / SHA-384 beside the SHA-256 compression of ecgpu_sha256.h (SHA-224 = SHA-256 with its own
// initial value).  The reference takes these from the un-vendored crates sha2 / sm3 (Cargo.lock) and binds one digest to each
// curve (`DigestAlgorithm`: k256/src/ecdsa.rs:117-119, p256/src/ecdsa.rs:72-74 Sha256; p384/src/ecdsa.rs:69-71 Sha384;
// p224/src/ecdsa.rs:69-71 Sha224; p521/src/ecdsa.rs:69-71 Sha512; bp256 / bp384 likewise); FIPS 180-4 is restated here and
// pinned by hashlib in the tests and by the reference's Wycheproof vectors at the message level.
// One message per lane; the EC work that follows is 100x larger.
#pragma once

#include <cstddef>
#include <cstdint>

#include "ecgpu_params.h"
#include "ecgpu_sha256.h"

namespace ecgpu {

struct HashPiece {
    const uint8_t* p;
    size_t n;
};

// Core: word_t, BLOCK_BYTES (16 words), LEN_BYTES (length field), compress(state, 16 big-endian words).
// state holds the initial value on entry and the final chaining value on return.
// The fully unrolled compression function is inlined exactly once per instantiation (a `put(byte)`-style absorber inlined
// it at every call site and the first SM2 kernel did not finish compiling).
template <class Core, int NP>
ECGPU_HD void hash_pieces(typename Core::word_t* state, const HashPiece* pc) {
    using W = typename Core::word_t;
    constexpr int WB = (int)sizeof(W), BB = Core::BLOCK_BYTES, LB = Core::LEN_BYTES;
    size_t total = 0;
#pragma unroll
    for (int t = 0; t < NP; t++) total += pc[t].n;
    const uint64_t bits = (uint64_t)total * 8;                   // < 2^64: the upper length bytes of SHA-512 stay zero
    const size_t nblocks = (total + 1 + LB + BB - 1) / BB;
    int cur = 0;                       // piece the cursor is in
    const uint8_t* cp = pc[0].p;       // its bytes and how many are left
    size_t left = pc[0].n;
    size_t o = 0;                      // offset in the padded message
#pragma unroll 1
    for (size_t blk = 0; blk < nblocks; blk++) {
        W w[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            W word = 0;
            if (o + WB <= total && left >= (size_t)WB && (reinterpret_cast<uintptr_t>(cp) & 3u) == 0) {
                // a whole word of message from the piece the cursor is in, 4-byte aligned: word loads (a 256-byte message
                // costs 0.6 instead of 2.3 ms per 2^20 this way); everything else goes byte by byte below
                const uint32_t* cw = static_cast<const uint32_t*>(__builtin_assume_aligned(cp, 4));
#pragma unroll
                for (int q = 0; q < WB / 4; q++) word = (W)((W)(word << 16) << 16) | (W)bswap32(cw[q]);
                cp += WB;
                left -= WB;
                o += WB;
                w[j] = word;
                continue;
            }
#pragma unroll 1
            for (int k = 0; k < WB; k++, o++) {
                uint32_t byte = 0;
                if (o < total) {
                    while (left == 0) {                           // next non-empty piece (there is one: o < total)
                        cur++;
#pragma unroll
                        for (int t = 1; t < NP; t++) {
                            if (t == cur) { cp = pc[t].p; left = pc[t].n; }
                        }
                    }
                    byte = *cp++;
                    left--;
                } else if (o == total) {
                    byte = 0x80u;
                } else if (o >= nblocks * BB - 8) {
                    byte = (uint32_t)(bits >> (8 * (nblocks * BB - 1 - o))) & 0xffu;
                }
                word = (W)(word << 8) | (W)byte;
            }
            w[j] = word;
        }
        Core::compress(state, w);
    }
}

struct Sha256Core {
    using word_t = uint32_t;
    ECGPU_CONST int BLOCK_BYTES = 64, LEN_BYTES = 8;
    static ECGPU_HD void compress(uint32_t* h, const uint32_t* w) { Sha256::compress(h, w); }
    static ECGPU_HD void init256(uint32_t* h) {
        h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
        h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
    }
    static ECGPU_HD void init224(uint32_t* h) {
        const uint32_t iv[8] = {
        0xc1059ed8u, 0x367cd507u, 0x3070dd17u, 0xf70e5939u, 0xffc00b31u, 0x68581511u, 0x64f98fa7u, 0xbefa4fa4u};
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = iv[i];
    }
};

struct Sha512Core {
    using word_t = uint64_t;
    ECGPU_CONST int BLOCK_BYTES = 128, LEN_BYTES = 16;
    ECGPU_CONST uint64_t K[80] = {
        0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull,
        0x3956c25bf348b538ull, 0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull,
        0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
        0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull,
        0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
        0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
        0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull,
        0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull, 0x06ca6351e003826full, 0x142929670a0e6e70ull,
        0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
        0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
        0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull,
        0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
        0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull,
        0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull,
        0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
        0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull,
        0xca273eceea26619cull, 0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull,
        0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
        0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull,
        0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
    static ECGPU_HD uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    static ECGPU_HD void compress(uint64_t* h, const uint64_t* block) {
        uint64_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = block[i];
        uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 8
        for (int i = 0; i < 80; i++) {
            if (i >= 16) {
                const uint64_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint64_t s0 = rotr(w15, 1) ^ rotr(w15, 8) ^ (w15 >> 7);
                const uint64_t s1 = rotr(w2, 19) ^ rotr(w2, 61) ^ (w2 >> 6);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint64_t S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41);
            const uint64_t ch = (e & f) ^ (~e & g);
            const uint64_t t1 = hh + S1 + ch + K[i] + w[i & 15];
            const uint64_t S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
            const uint64_t maj = (a & b) ^ (a & c) ^ (b & c);
            const uint64_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    static ECGPU_HD void init512(uint64_t* h) {
        const uint64_t iv[8] = {
        0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
        0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = iv[i];
    }
    static ECGPU_HD void init384(uint64_t* h) {
        const uint64_t iv[8] = {
        0xcbbb9d5dc1059ed8ull, 0x629a292a367cd507ull, 0x9159015a3070dd17ull, 0x152fecd8f70e5939ull,
        0x67332667ffc00b31ull, 0x8eb44a8768581511ull, 0xdb0c2e0d64f98fa7ull, 0x47b5481dbefa4fa4ull};
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = iv[i];
    }
};

// digest ids (also the digest length in bytes)
enum : int { HASH_SHA224 = 28, HASH_SHA256 = 32, HASH_SHA384 = 48, HASH_SHA512 = 64 };

// SHA-2 of the concatenation of NP byte strings: DIGEST bytes into out
template <int DIGEST, int NP>
ECGPU_HD void sha2_pieces(uint8_t* out, const HashPiece* pc) {
    if constexpr (DIGEST == HASH_SHA224 || DIGEST == HASH_SHA256) {
        uint32_t h[8];
        if constexpr (DIGEST == HASH_SHA224) Sha256Core::init224(h); else Sha256Core::init256(h);
        hash_pieces<Sha256Core, NP>(h, pc);
#pragma unroll
        for (int i = 0; i < DIGEST / 4; i++) {
            out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
        }
    } else {
        static_assert(DIGEST == HASH_SHA384 || DIGEST == HASH_SHA512, "SHA-224 / 256 / 384 / 512");
        uint64_t h[8];
        if constexpr (DIGEST == HASH_SHA384) Sha512Core::init384(h); else Sha512Core::init512(h);
        hash_pieces<Sha512Core, NP>(h, pc);
#pragma unroll
        for (int i = 0; i < DIGEST / 8; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(h[i] >> (8 * (7 - k)));
        }
    }
}

// The digest the reference binds to a curve (`DigestAlgorithm`); 0 = none (p192 has no impl, sm2 / bign are not ECDSA)
template <class C>
struct EcdsaDigest {
    ECGPU_CONST int value = C::ID == CURVE_K256 || C::ID == CURVE_P256 || C::ID == CURVE_BP256 || C::ID == CURVE_BP256T1 ? HASH_SHA256
                            : C::ID == CURVE_P384 || C::ID == CURVE_BP384 || C::ID == CURVE_BP384T1               ? HASH_SHA384
                            : C::ID == CURVE_P224                                                                  ? HASH_SHA224
                            : C::ID == CURVE_P521                                                                  ? HASH_SHA512
                                                                                                                   : 0;
};

}  // namespace ecgpu
