/* examples/batch_pubkeys.c — the C ABI from plain C: derive secp256k1 public keys for a batch of private keys
 * (`mul_by_generator`), then run the same batch through ECDH against one peer key.
 *
 *     make -C examples            # gcc, links ../elliptic-curves_amd/lib/libecgpu.so
 *     ./examples/batch_pubkeys    # needs an MI355X
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

int main(void) {
    ecgpu_ctx *ctx = NULL;
    CHECK(ecgpu_init(&ctx, 0));
    enum { N = 4096, L = 32 };
    uint8_t *priv = calloc(N, L), *pub = malloc(N * 2 * L), *inf = malloc(N);
    uint8_t *peer = malloc(N * 2 * L), *shared = malloc(N * L), *ok = malloc(N);
    for (int i = 0; i < N; i++) {                 /* private keys 1, 2, 3, ... (big-endian) */
        priv[i * L + L - 1] = (uint8_t)((i + 1) & 0xff);
        priv[i * L + L - 2] = (uint8_t)((i + 1) >> 8);
    }
    CHECK(ecgpu_batch_mul_base(ctx, ECGPU_K256, priv, N, pub, inf));
    printf("pub[0] = G:  x = ");
    for (int j = 0; j < L; j++) printf("%02x", pub[j]);
    printf("\n");
    for (int i = 0; i < N; i++) memcpy(peer + i * 2 * L, pub + 2 * L * 6, 2 * L);   /* everyone talks to key #7 */
    CHECK(ecgpu_batch_ecdh(ctx, ECGPU_K256, priv, peer, N, shared, ok));
    /* ECDH is symmetric: key 7's secret with key 1's public point equals key 1's secret with key 7's */
    uint8_t back[L], ok1;
    CHECK(ecgpu_batch_ecdh(ctx, ECGPU_K256, priv + 6 * L, pub, 1, back, &ok1));
    printf("shared(1,7) == shared(7,1): %s\n", memcmp(back, shared, L) == 0 && ok1 && ok[0] ? "yes" : "NO");
    ecgpu_destroy(ctx);
    free(priv); free(pub); free(inf); free(peer); free(shared); free(ok);
    return 0;
}
