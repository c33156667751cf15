// ecgpu_sha256.h — SHA-256 compression and the BIP340 challenge hash (host + device).
//
// Only `schnorr::VerifyingKey::verify_raw` needs it (k256/src/schnorr/verifying.rs:79-85 and the `tagged_hash` helper of
// k256/src/schnorr.rs): e = SHA256(SHA256(tag) || SHA256(tag) || r || pk || m) with tag = "BIP0340/challenge".  The
// first 64-byte block (tag hash twice) is the same for every signature, so hashing starts from its midstate.
// FIPS 180-4; one message per lane, the EC work that follows is 100x larger.
#pragma once

#include <cstddef>
#include <cstdint>

#include "ecgpu_params.h"

namespace ecgpu {

struct Sha256 {
    ECGPU_CONST uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    // state after the block SHA256("BIP0340/challenge") || SHA256("BIP0340/challenge")   (tools: hashlib)
    ECGPU_CONST uint32_t BIP340_CHALLENGE_MIDSTATE[8] = {0x9CECBA11u, 0x23925381u, 0x11679112u, 0xD1627E0Fu,
                                                         0x97C87550u, 0x003CC765u, 0x90F61164u, 0x33E9B66Au};

    static ECGPU_HD uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

    // h <- compress(h, block); block as 16 big-endian words
    static ECGPU_HD void compress(uint32_t* h, const uint32_t* block) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = block[i];
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; i++) {
            if (i >= 16) {
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            const uint32_t ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            const uint32_t maj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }

    // e = SHA256(tag || tag || r || pk || m) as 8 LITTLE-endian words of the big-endian 256-bit integer
    // (ready for the scalar arithmetic).  r, pk: 32 bytes each; m: msg_len bytes.
    static ECGPU_HD void bip340_challenge(uint32_t* e_words, const uint8_t* r, const uint8_t* pk, const uint8_t* m,
                                          size_t msg_len) {
        uint32_t h[8];
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = BIP340_CHALLENGE_MIDSTATE[i];
        const size_t total = 64 + msg_len;                       // bytes hashed after the midstate block
        const uint64_t bitlen = (uint64_t)(64 + total) * 8;      // whole message incl. the tag block
        const size_t nblocks = (total + 9 + 63) / 64;
        for (size_t blk = 0; blk < nblocks; blk++) {
            uint32_t w[16];
            for (int j = 0; j < 16; j++) {
                uint32_t word = 0;
                for (int k = 0; k < 4; k++) {
                    const size_t o = blk * 64 + (size_t)j * 4 + k;
                    uint32_t byte;
                    if (o < 32) byte = r[o];
                    else if (o < 64) byte = pk[o - 32];
                    else if (o < total) byte = m[o - 64];
                    else if (o == total) byte = 0x80;
                    else if (o >= nblocks * 64 - 8) byte = (uint32_t)(bitlen >> (8 * (nblocks * 64 - 1 - o))) & 0xff;
                    else byte = 0;
                    word = (word << 8) | byte;
                }
                w[j] = word;
            }
            compress(h, w);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) e_words[i] = h[7 - i];
    }
};

}  // namespace ecgpu
