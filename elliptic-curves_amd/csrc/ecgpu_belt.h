// ecgpu_belt.h — belt-hash (STB 34.101.31-2020 §7.8 over the belt block cipher, §7.1-7.2) for bign verification (host + device).
//
// `bignp256::ecdsa::VerifyingKey` hashes twice with it: H = belt-hash(message) (`hash_msg`, bignp256/src/ecdsa/verifying.rs:87-91)
// and t = belt-hash(OID(h) || <R>_2l || H) inside `verify_prehash` (:131-139).  The hash lives in the un-vendored crates
// `belt-hash` / `belt-block` (bignp256/Cargo.toml); the standard's published algorithm is restated here and pinned end to end by
// the reference's own signature vector (bignp256/tests/ecdsa.rs:21-46, from STB 34.101.45 via met-10145): a valid signature only
// verifies if every byte of the S-box and every step below is right.  One message per lane; the EC work that follows is 100x larger.
#pragma once

#include <cstddef>
#include <cstdint>

#include "ecgpu_hash.h"
#include "ecgpu_params.h"

namespace ecgpu {

struct Belt {
    // the S-box H (STB 34.101.31 table 1); its first 32 bytes are also the initial value of belt-hash
    ECGPU_CONST uint8_t H[256] = {
        0xB1, 0x94, 0xBA, 0xC8, 0x0A, 0x08, 0xF5, 0x3B, 0x36, 0x6D, 0x00, 0x8E, 0x58, 0x4A, 0x5D, 0xE4,
        0x85, 0x04, 0xFA, 0x9D, 0x1B, 0xB6, 0xC7, 0xAC, 0x25, 0x2E, 0x72, 0xC2, 0x02, 0xFD, 0xCE, 0x0D,
        0x5B, 0xE3, 0xD6, 0x12, 0x17, 0xB9, 0x61, 0x81, 0xFE, 0x67, 0x86, 0xAD, 0x71, 0x6B, 0x89, 0x0B,
        0x5C, 0xB0, 0xC0, 0xFF, 0x33, 0xC3, 0x56, 0xB8, 0x35, 0xC4, 0x05, 0xAE, 0xD8, 0xE0, 0x7F, 0x99,
        0xE1, 0x2B, 0xDC, 0x1A, 0xE2, 0x82, 0x57, 0xEC, 0x70, 0x3F, 0xCC, 0xF0, 0x95, 0xEE, 0x8D, 0xF1,
        0xC1, 0xAB, 0x76, 0x38, 0x9F, 0xE6, 0x78, 0xCA, 0xF7, 0xC6, 0xF8, 0x60, 0xD5, 0xBB, 0x9C, 0x4F,
        0xF3, 0x3C, 0x65, 0x7B, 0x63, 0x7C, 0x30, 0x6A, 0xDD, 0x4E, 0xA7, 0x79, 0x9E, 0xB2, 0x3D, 0x31,
        0x3E, 0x98, 0xB5, 0x6E, 0x27, 0xD3, 0xBC, 0xCF, 0x59, 0x1E, 0x18, 0x1F, 0x4C, 0x5A, 0xB7, 0x93,
        0xE9, 0xDE, 0xE7, 0x2C, 0x8F, 0x0C, 0x0F, 0xA6, 0x2D, 0xDB, 0x49, 0xF4, 0x6F, 0x73, 0x96, 0x47,
        0x06, 0x07, 0x53, 0x16, 0xED, 0x24, 0x7A, 0x37, 0x39, 0xCB, 0xA3, 0x83, 0x03, 0xA9, 0x8B, 0xF6,
        0x92, 0xBD, 0x9B, 0x1C, 0xE5, 0xD1, 0x41, 0x01, 0x54, 0x45, 0xFB, 0xC9, 0x5E, 0x4D, 0x0E, 0xF2,
        0x68, 0x20, 0x80, 0xAA, 0x22, 0x7D, 0x64, 0x2F, 0x26, 0x87, 0xF9, 0x34, 0x90, 0x40, 0x55, 0x11,
        0xBE, 0x32, 0x97, 0x13, 0x43, 0xFC, 0x9A, 0x48, 0xA0, 0x2A, 0x88, 0x5F, 0x19, 0x4B, 0x09, 0xA1,
        0x7E, 0xCD, 0xA4, 0xD0, 0x15, 0x44, 0xAF, 0x8C, 0xA5, 0x84, 0x50, 0xBF, 0x66, 0xD2, 0xE8, 0x8A,
        0xA2, 0xD7, 0x46, 0x52, 0x42, 0xA8, 0xDF, 0xB3, 0x69, 0x74, 0xC5, 0x51, 0xEB, 0x23, 0x29, 0x21,
        0xD4, 0xEF, 0xD9, 0xB4, 0x3A, 0x62, 0x28, 0x75, 0x91, 0x14, 0x10, 0xEA, 0x77, 0x6C, 0xDA, 0x1D};
    // DER of the object identifier of belt-hash, 1.2.112.0.2.0.34.101.31.81 (`BELT_OID`, bignp256/src/ecdsa.rs:58-60)
    ECGPU_CONST uint8_t OID[11] = {0x06, 0x09, 0x2A, 0x70, 0x00, 0x02, 0x00, 0x22, 0x65, 0x1F, 0x51};

    // G_r: the S-box on the four bytes of a word, then a left rotation by r.  `sb` = the table the caller wants the 2,700 lookups
    // of one signature to go to: Belt::H itself on the host, a copy in LDS in the kernels (normalise + k_bign_finish per
    // 2^20 signatures: 1.60 -> 0.44 ms, profiles/r03/bign_verify_rate.txt).
    template <int R>
    static ECGPU_HD uint32_t g(const uint8_t* sb, uint32_t u) {
        const uint32_t v = (uint32_t)sb[u & 255u] | (uint32_t)sb[(u >> 8) & 255u] << 8 | (uint32_t)sb[(u >> 16) & 255u] << 16 |
                           (uint32_t)sb[u >> 24] << 24;
        return (v << R) | (v >> (32 - R));
    }

    // y <- belt-block(x) under the 256-bit key: eight rounds over the words a, b, c, d (all words little-endian), key words
    // K[7i - 6 .. 7i] = key[(7i - 7 .. 7i - 1) mod 8]
    static ECGPU_HD void block(const uint8_t* sb, uint32_t* y, const uint32_t* x, const uint32_t* key) {
        uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
#pragma unroll
        for (int i = 1; i <= 8; i++) {
            const int k0 = 7 * (i - 1);
            b ^= g<5>(sb, a + key[(k0 + 0) & 7]);
            c ^= g<21>(sb, d + key[(k0 + 1) & 7]);
            a -= g<13>(sb, b + key[(k0 + 2) & 7]);
            const uint32_t e = g<21>(sb, b + c + key[(k0 + 3) & 7]) ^ (uint32_t)i;
            b += e;
            c -= e;
            d += g<13>(sb, c + key[(k0 + 4) & 7]);
            b ^= g<21>(sb, a + key[(k0 + 5) & 7]);
            c ^= g<5>(sb, d + key[(k0 + 6) & 7]);
            uint32_t t = a; a = b; b = t;       // a <-> b
            t = c; c = d; d = t;                // c <-> d
            t = b; b = c; c = t;                // b <-> c
        }
        y[0] = b; y[1] = d; y[2] = a; y[3] = c;
    }

    // One step of belt-hash on u = x (8 words) || h (8 words):
    //     s <- s xor sigma1(u),   h <- sigma2(u)
    // sigma1(u) = belt-block(u3 xor u4, u1 || u2) xor u3 xor u4;
    // sigma2(u) = (belt-block(u1, sigma1(u) || u4) xor u1) || (belt-block(u2, (sigma1(u) xor 1^128) || u3) xor u2)
    static ECGPU_HD void step(const uint8_t* sb, uint32_t* s, uint32_t* h, const uint32_t* x) {
        uint32_t t[4], s1[4], key[8], y1[4], y2[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = h[j] ^ h[4 + j];
        block(sb, s1, t, x);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s1[j] ^= t[j];
            s[j] ^= s1[j];
            key[j] = s1[j];
            key[4 + j] = h[4 + j];
        }
        block(sb, y1, x, key);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            key[j] = ~s1[j];
            key[4 + j] = h[j];
        }
        block(sb, y2, x + 4, key);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            h[j] = y1[j] ^ x[j];
            h[4 + j] = y2[j] ^ x[4 + j];
        }
    }

    // out (8 little-endian words = the 32 digest bytes in order) = belt-hash(pieces[0] || ... || pieces[NP - 1]).
    // Blocks of 32 bytes, the last one filled with zeros; then one more step on <bit length>_128 || s.  ONE call site of `step`
    // (the last pass of the loop is the finalisation), so the three unrolled block encryptions are inlined once.
    template <int NP>
    static ECGPU_HD void hash_pieces(const uint8_t* sb, uint32_t* out, const HashPiece* pc) {
        size_t total = 0;
#pragma unroll
        for (int t = 0; t < NP; t++) total += pc[t].n;
        const size_t nblocks = (total + 31) / 32;
        uint32_t s[4] = {0, 0, 0, 0}, h[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            h[j] = (uint32_t)H[4 * j] | (uint32_t)H[4 * j + 1] << 8 | (uint32_t)H[4 * j + 2] << 16 | (uint32_t)H[4 * j + 3] << 24;
        int cur = 0;
        const uint8_t* cp = pc[0].p;
        size_t left = pc[0].n, o = 0;
#pragma unroll 1
        for (size_t blk = 0; blk <= nblocks; blk++) {
            uint32_t x[8];
            if (blk == nblocks) {
                const uint64_t bits = (uint64_t)total * 8;
                x[0] = (uint32_t)bits; x[1] = (uint32_t)(bits >> 32); x[2] = 0; x[3] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) x[4 + j] = s[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    uint32_t word = 0;
#pragma unroll 1
                    for (int k = 0; k < 4; k++, o++) {
                        uint32_t byte = 0;
                        if (o < total) {
                            while (left == 0) {                   // next non-empty piece (there is one: o < total)
                                cur++;
#pragma unroll
                                for (int t = 1; t < NP; t++) {
                                    if (t == cur) { cp = pc[t].p; left = pc[t].n; }
                                }
                            }
                            byte = *cp++;
                            left--;
                        }
                        word |= byte << (8 * k);
                    }
                    x[j] = word;
                }
            }
            step(sb, s, h, x);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) out[j] = h[j];
    }
};

}  // namespace ecgpu
