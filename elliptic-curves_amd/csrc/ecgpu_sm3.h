// ecgpu_sm3.h — SM3 (GB/T 32905-2016) and the two hashes SM2DSA verification from wire bytes needs (host + device).
//
// `sm2::dsa::VerifyingKey::new(distid, public_key)` computes the signer's identity hash
//     Z = SM3(ENTL || ID || a || b || xG || yG || xA || yA)          (`hash_z`, sm2/src/distid.rs:21-44; ENTL = the bit
//                                                                     length of ID as 2 big-endian bytes)
// and `Verifier::verify(msg, sig)` hashes e = SM3(Z || M) (`hash_msg`, sm2/src/dsa/verifying.rs:126-130) before
// `verify_prehash` (:138-171).  The compression function itself lives in the un-vendored crate `sm3` 0.5 (Cargo.lock);
// its published algorithm is restated here and pinned through the reference's message-level vector (sm2/tests/sm2dsa.rs:16-35)
// and OpenSSL's SM3 (hashlib) in the tests.  One message per lane; the EC work that follows is 100x larger.
#pragma once

#include <cstddef>
#include <cstdint>

#include "ecgpu_hash.h"
#include "ecgpu_params.h"

namespace ecgpu {

struct Sm3 {
    static ECGPU_HD uint32_t rotl(uint32_t x, int n) {
        n &= 31;
        return n ? (x << n) | (x >> (32 - n)) : x;
    }
    static ECGPU_HD uint32_t p0(uint32_t x) { return x ^ rotl(x, 9) ^ rotl(x, 17); }
    static ECGPU_HD uint32_t p1(uint32_t x) { return x ^ rotl(x, 15) ^ rotl(x, 23); }

    static ECGPU_HD void init(uint32_t* v) {
        v[0] = 0x7380166fu; v[1] = 0x4914b2b9u; v[2] = 0x172442d7u; v[3] = 0xda8a0600u;
        v[4] = 0xa96f30bcu; v[5] = 0x163138aau; v[6] = 0xe38dee4du; v[7] = 0xb0fb0e4eu;
    }
    // v <- CF(v, block); block as 16 big-endian words
    static ECGPU_HD void compress(uint32_t* v, const uint32_t* block) {
        uint32_t w[68];
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = block[j];
#pragma unroll
        for (int j = 16; j < 68; j++) w[j] = p1(w[j - 16] ^ w[j - 9] ^ rotl(w[j - 3], 15)) ^ rotl(w[j - 13], 7) ^ w[j - 6];
        uint32_t a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6], h = v[7];
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const uint32_t t = j < 16 ? 0x79cc4519u : 0x7a879d8au;
            const uint32_t a12 = rotl(a, 12);
            const uint32_t ss1 = rotl(a12 + e + rotl(t, j), 7);
            const uint32_t ss2 = ss1 ^ a12;
            const uint32_t ff = j < 16 ? (a ^ b ^ c) : ((a & b) | (a & c) | (b & c));
            const uint32_t gg = j < 16 ? (e ^ f ^ g) : ((e & f) | (~e & g));
            const uint32_t tt1 = ff + d + ss2 + (w[j] ^ w[j + 4]);
            const uint32_t tt2 = gg + h + ss1 + w[j];
            d = c; c = rotl(b, 9); b = a; a = tt1;
            h = g; g = rotl(f, 19); f = e; e = p0(tt2);
        }
        v[0] ^= a; v[1] ^= b; v[2] ^= c; v[3] ^= d; v[4] ^= e; v[5] ^= f; v[6] ^= g; v[7] ^= h;
    }

    // hash_pieces<Sm3, NP> (ecgpu_hash.h) drives `compress` over a concatenation of byte strings
    using word_t = uint32_t;
    ECGPU_CONST int BLOCK_BYTES = 64, LEN_BYTES = 8;

    // a || b || xG || yG of the sm2 curve as the 32-byte big-endian strings `to_bytes()` yields (sm2/src/arithmetic.rs:53-74);
    // `sm2_constants_match` ties them to the parameter pack at compile time
    ECGPU_CONST uint8_t SM2_ABG[128] = {
        0xFF, 0xFF, 0xFF, 0xFE, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF,
        0xFF, 0xFF, 0xFF, 0xFF, 0x00, 0x00, 0x00, 0x00, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFC,
        0x28, 0xE9, 0xFA, 0x9E, 0x9D, 0x9F, 0x5E, 0x34, 0x4D, 0x5A, 0x9E, 0x4B, 0xCF, 0x65, 0x09, 0xA7,
        0xF3, 0x97, 0x89, 0xF5, 0x15, 0xAB, 0x8F, 0x92, 0xDD, 0xBC, 0xBD, 0x41, 0x4D, 0x94, 0x0E, 0x93,
        0x32, 0xC4, 0xAE, 0x2C, 0x1F, 0x19, 0x81, 0x19, 0x5F, 0x99, 0x04, 0x46, 0x6A, 0x39, 0xC9, 0x94,
        0x8F, 0xE3, 0x0B, 0xBF, 0xF2, 0x66, 0x0B, 0xE1, 0x71, 0x5A, 0x45, 0x89, 0x33, 0x4C, 0x74, 0xC7,
        0xBC, 0x37, 0x36, 0xA2, 0xF4, 0xF6, 0x77, 0x9C, 0x59, 0xBD, 0xCE, 0xE3, 0x6B, 0x69, 0x21, 0x53,
        0xD0, 0xA9, 0x87, 0x7C, 0xC6, 0x2A, 0x47, 0x40, 0x02, 0xDF, 0x32, 0xE5, 0x21, 0x39, 0xF0, 0xA0};
    template <class C>
    static constexpr bool sm2_constants_match() {
        for (int i = 0; i < 8; i++) {
            for (int k = 0; k < 4; k++) {
                const int at = 4 * (7 - i) + (3 - k);             // byte of little-endian word i, bits 8k..8k+7
                const uint32_t a = (C::P[i] - (i == 0 ? 3u : 0u)) >> (8 * k) & 0xffu;
                if (SM2_ABG[at] != a || SM2_ABG[32 + at] != (C::B[i] >> (8 * k) & 0xffu) ||
                    SM2_ABG[64 + at] != (C::GX[i] >> (8 * k) & 0xffu) || SM2_ABG[96 + at] != (C::GY[i] >> (8 * k) & 0xffu))
                    return false;
            }
        }
        return true;
    }

    // e = SM3(Z || M) with Z = SM3(ENTL || ID || a || b || xG || yG || xA || yA), as 8 LITTLE-endian words of the big-endian
    // 256-bit integer (ready for `Scalar::reduce`); q_xy = the key's 64 wire bytes.
    template <class C>
    static ECGPU_HD void sm2_message_hash(uint32_t* e_words, const uint8_t* distid, size_t distid_len, const uint8_t* q_xy,
                                          const uint8_t* msg, size_t msg_len) {
        static_assert(sm2_constants_match<C>(), "SM2_ABG does not restate the curve's a, b and generator");
        const uint32_t entl_bits = (uint32_t)(distid_len * 8);
        const uint8_t entl[2] = {(uint8_t)(entl_bits >> 8), (uint8_t)entl_bits};
        const HashPiece zin[4] = {{entl, 2}, {distid, distid_len}, {SM2_ABG, 128}, {q_xy, 64}};
        uint32_t z[8];
        init(z);
        hash_pieces<Sm3, 4>(z, zin);
        uint8_t zb[32];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            zb[4 * i] = (uint8_t)(z[i] >> 24); zb[4 * i + 1] = (uint8_t)(z[i] >> 16); zb[4 * i + 2] = (uint8_t)(z[i] >> 8); zb[4 * i + 3] = (uint8_t)z[i];
        }
        const HashPiece ein[2] = {{zb, 32}, {msg, msg_len}};
        uint32_t d[8];
        init(d);
        hash_pieces<Sm3, 2>(d, ein);
#pragma unroll
        for (int i = 0; i < 8; i++) e_words[i] = d[7 - i];
    }
    // SM3 of one byte string (test hook)
    static ECGPU_HD void hash(uint32_t* digest, const uint8_t* msg, size_t len) {
        const HashPiece one[1] = {{msg, len}};
        init(digest);
        hash_pieces<Sm3, 1>(digest, one);
    }
};

}  // namespace ecgpu
