/* examples/verify_and_recover.c — the signature entry points of the C ABI from plain C, on the reference's own vectors:
 *   - public-key recovery (`VerifyingKey::recover_from_prehash`): k256/src/ecdsa.rs:190-211 RECOVERY_TEST_VECTORS —
 *     "example message", SHA-256 digest (hard-coded below), recovery ids 0 and 1, the expected keys SEC1-compressed;
 *   - message-level ECDSA verification (`Verifier::verify`): the same two signatures against the recovered keys, the digest
 *     computed on the device from the message bytes;
 *   - message-level SM2DSA verification (`VerifyingKey::new(distid, pk)?.verify(msg, sig)`): sm2/tests/sm2dsa.rs:16-31.
 *
 *     make -C examples && ./examples/verify_and_recover      # needs an MI355X
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ecgpu.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != ECGPU_OK) {                                                                   \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? ecgpu_last_error(ctx) : ""); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

void example_sha256(uint8_t out[32], const uint8_t *msg, size_t len);

static void unhex(uint8_t *out, const char *hex) {
    for (size_t i = 0; hex[2 * i]; i++) {
        unsigned v;
        sscanf(hex + 2 * i, "%2x", &v);
        out[i] = (uint8_t)v;
    }
}

int main(void) {
    ecgpu_ctx *ctx = NULL;
    CHECK(ecgpu_init(&ctx, 0));
    enum { L = 32 };
    static const char *MSG = "example message";
    static const char *SIG[2] = {
        "ce53abb3721bafc561408ce8ff99c909f7f0b18a2f788649d6470162ab1aa0323971edc523a6d6453f3fb6128d318d9db1a5ff3386feb1047d9816e780039d52",
        "46c05b6368a44b8810d79859441d819b8e7cdc8bfd371e35c53196f4bcacdb5135c7facce2a97b95eacba8a586d87b7958aaf8368ab29cee481f76e871dbd9cb"};
    static const char *PK[2] = {"021a7a569e91dbf60581509c7fc946d1003b60c7dee85299538db6353538d59574",
                                "036d6caac248af96f6afa7f904f550253a0f3ef3f5aa2fe6838a95b216691468e2"};
    uint8_t z[2 * L], r[2 * L], s[2 * L], sig[2 * 2 * L], recid[2] = {0, 1}, keys[2 * 2 * L], ok[2], pk[33];
    int good = 1;
    for (int i = 0; i < 2; i++) {
        unhex(sig + i * 2 * L, SIG[i]);
        memcpy(r + i * L, sig + i * 2 * L, L);
        memcpy(s + i * L, sig + i * 2 * L + L, L);
    }
    /* the prehash entry point takes the digest the caller computed: SHA-256 of the message (helper at the end of the file) */
    example_sha256(z, (const uint8_t *)MSG, strlen(MSG));
    memcpy(z + L, z, L);
    CHECK(ecgpu_ecdsa_recover_batch(ctx, ECGPU_K256, z, r, s, recid, 2, /* NORMALIZE_S */ 1, keys, ok));
    for (int i = 0; i < 2; i++) {
        unhex(pk, PK[i]);
        const int match = ok[i] && memcmp(keys + i * 2 * L, pk + 1, L) == 0 && (keys[i * 2 * L + 2 * L - 1] & 1) == (pk[0] & 1);
        printf("recovered key %d == the reference's: %s\n", i, match ? "yes" : "NO");
        good &= match;
    }
    /* message-level verification of the same signatures under the recovered keys: the digest is computed on the device */
    {
        uint8_t msgs[2 * 15], vok[2];
        memcpy(msgs, MSG, 15);
        memcpy(msgs + 15, MSG, 15);
        CHECK(ecgpu_ecdsa_verify_msg_batch(ctx, ECGPU_K256, keys, msgs, 15, sig, 2, 1, vok));
        printf("Verifier::verify(msg, sig) under the recovered keys: %s\n", vok[0] && vok[1] ? "yes" : "NO");
        good &= vok[0] && vok[1];
        msgs[0] ^= 1;
        CHECK(ecgpu_ecdsa_verify_msg_batch(ctx, ECGPU_K256, keys, msgs, 15, sig, 2, 1, vok));
        printf("a disturbed message is rejected: %s\n", !vok[0] && vok[1] ? "yes" : "NO");
        good &= !vok[0] && vok[1];
    }
    /* SM2DSA: sm2/tests/sm2dsa.rs:16-31 */
    {
        uint8_t q[65], sg[64], vok = 0;
        unhex(q, "0408D77AE04C01CC4C1104360DD8AF6B6F7DF334283D7C1A6AFD5652407B87BEE5014E2A57C36C150D16324DC664E31E6432359609C4E79847A5B161C8C7364C8A");
        unhex(sg, "d1dcccedd9fb785e0f67c16b7c52901625c0b69de9bca2144acc7be713cad2fcf7d1eae6e3a157b36c65f672f738ca8b46298bf149a6510072c431b49cd88b1c");
        static const char *ID = "example@rustcrypto.org";
        /* host buffers may have any alignment: q + 1 skips the SEC1 tag byte */
        CHECK(ecgpu_sm2dsa_verify_msg_batch(ctx, (const uint8_t *)ID, strlen(ID), q + 1, (const uint8_t *)"testing", 7, sg, 1, &vok));
        printf("SM2DSA verify(\"testing\") for %s: %s\n", ID, vok ? "yes" : "NO");
        good &= vok;
    }
    ecgpu_destroy(ctx);
    return good ? 0 : 2;
}

/* ---- a small SHA-256 so that the example needs nothing but libc (FIPS 180-4) ---------------------------------------- */
static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void example_sha256(uint8_t out[32], const uint8_t *msg, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[128] = {0};
    if (len > 55) return;                                        /* one block is all this example needs */
    memcpy(buf, msg, len);
    buf[len] = 0x80;
    buf[62] = (uint8_t)((len * 8) >> 8);
    buf[63] = (uint8_t)(len * 8);
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)buf[4 * i] << 24) | ((uint32_t)buf[4 * i + 1] << 16) | ((uint32_t)buf[4 * i + 2] << 8) | buf[4 * i + 3];
    for (int i = 16; i < 64; i++)
        w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
}
