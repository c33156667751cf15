/* examples/node_lincomb.c — `lincomb` over every GPU of a node from ONE process, in plain C (the ecgpu_group_* entry points
 * of include/ecgpu.h; what the Rust caller of `LinearCombination::lincomb_vartime` would do through ecgpu_shim.rs).
 *
 *     make -C examples
 *     ./examples/node_lincomb [ndev] [log2 terms] [peer|rccl]    # default: every visible GPU (at most 8), 2^18 terms, RCCL when it
 *                                                                # can be had (rccl: or fail; peer: peer copies) — ecgpu_group_set_exchange
 *
 * The terms are k_i * P_i with P_i = s_i * G made on the GPUs themselves; the check is the group identity
 *     lincomb over the whole node  ==  lincomb on GPU 0 alone,
 * byte for byte.  On a one-GPU box pass ndev = 2: the same device is then listed twice (two contexts, two worker
 * threads, the exchange by peer copy), which exercises the same code path.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ecgpu.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next64(void) {                       /* SplitMix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void random_scalars(uint8_t *out, size_t n) { /* 32 random bytes, top word kept below the group order's */
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < 4; j++) {
            uint64_t v = next64();
            memcpy(out + 32 * i + 8 * j, &v, 8);
        }
        out[32 * i] &= 0x7f;
    }
}

int main(int argc, char **argv) {
    int ndev = argc > 1 ? atoi(argv[1]) : 0;
    const int lg = argc > 2 ? atoi(argv[2]) : 18;
    const size_t n = (size_t)1 << lg;
    int devices[8];
    ecgpu_group *grp = NULL;
    if (ndev <= 0) {                                   /* as many distinct GPUs as the runtime shows (at most 8) */
        for (ndev = 8; ndev >= 1; ndev--) {
            for (int i = 0; i < ndev; i++) devices[i] = i;
            if (ecgpu_group_init(&grp, devices, ndev) == ECGPU_OK) break;
        }
        if (!grp) { fprintf(stderr, "no gfx950 device\n"); return 1; }
    } else {
        if (ndev > 8) ndev = 8;
        /* distinct devices if there are that many, otherwise device 0 listed ndev times */
        for (int i = 0; i < ndev; i++) devices[i] = i;
        if (ecgpu_group_init(&grp, devices, ndev) != ECGPU_OK) {
            for (int i = 0; i < ndev; i++) devices[i] = 0;
            if (ecgpu_group_init(&grp, devices, ndev) != ECGPU_OK) { fprintf(stderr, "no gfx950 device\n"); return 1; }
        }
    }
    if (argc > 3 && ecgpu_group_set_exchange(grp, strcmp(argv[3], "peer") == 0 ? ECGPU_EXCHANGE_PEER : ECGPU_EXCHANGE_RCCL) != ECGPU_OK) {
        fprintf(stderr, "no gfx950 device with that exchange: %s\n", ecgpu_group_last_error(grp));
        return 1;
    }
    printf("group of %d member(s), exchange by %s, %zu terms\n", ecgpu_group_size(grp), ecgpu_group_exchange(grp), n);
    uint8_t *k = malloc(n * 32), *s = malloc(n * 32), *pts = malloc(n * 64);
    uint8_t node[64], one[64], ninf = 0, oinf = 0;
    random_scalars(k, n);
    random_scalars(s, n);
    int rc = ecgpu_group_batch_mul_base(grp, ECGPU_K256, s, n, pts, NULL);              /* P_i = s_i * G, sliced over the GPUs */
    if (rc == ECGPU_OK) rc = ecgpu_group_msm(grp, ECGPU_K256, k, pts, NULL, n, node, &ninf);
    if (rc != ECGPU_OK) { fprintf(stderr, "group call failed: %d (%s)\n", rc, ecgpu_group_last_error(grp)); return 1; }
    ecgpu_ctx *ctx0 = ecgpu_group_ctx(grp, 0);
    rc = ecgpu_msm(ctx0, ECGPU_K256, k, pts, NULL, n, one, &oinf);
    if (rc != ECGPU_OK) { fprintf(stderr, "ecgpu_msm failed: %d (%s)\n", rc, ecgpu_last_error(ctx0)); return 1; }
    printf("x = ");
    for (int j = 0; j < 32; j++) printf("%02x", node[j]);
    printf("\nnode lincomb == single-GPU lincomb: %s\n", memcmp(node, one, 64) == 0 && ninf == oinf ? "yes" : "NO");
    ecgpu_group_destroy(grp);
    free(k); free(s); free(pts);
    return memcmp(node, one, 64) == 0 ? 0 : 2;
}
