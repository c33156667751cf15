/*
 * ecgpu.h — C ABI of libecgpu.so, the MI355X (gfx950) batch scalar-multiplication / MSM engine
 * that drops in behind RustCrypto/elliptic-curves' scalar-mul surface for k256, p256 and p384.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * root).  The reference's own API works on one element at a time; the GPU entry points are the
 * batch forms of the same operations, plus `lincomb`, which is already a batch.  INTEGRATION.md
 * shows the Rust `extern "C"` block and the `MulBackend` / `LinearCombination` adapter a
 * maintainer would add on the reference side.
 *
 * Wire format (identical for host- and device-pointer entry points)
 *   scalars   n * L bytes, big-endian, canonical (< group order n) — `Scalar::to_repr`
 *             (k256/src/arithmetic/scalar.rs:310-316, primefield/src/monty.rs:498-500);
 *             L = 32 (k256, p256, sm2, bp256 = brainpoolP256r1, bp256t1, bign256), 48 (p384, bp384 = brainpoolP384r1, bp384t1), 28 (p224), 24 (p192) or 66 (p521) = `FieldBytesSize`.  Records are packed without
 *             padding (p224: 28-byte scalars, 56-byte points); only the base pointers of device buffers must be 16-byte aligned
 *             (which also gives every record the alignment its codec reads with: 4 bytes for p224 / p192, 2 bytes for p521's
 *             66-byte records — a *_dev pointer into the middle of an array must keep to a whole number of records).
 *   points    n * 2L bytes, big-endian affine x || y — `AffinePoint::{x,y}`
 *             (primeorder/src/affine.rs:106-112) + optional n-byte identity flags
 *             (`AffinePoint::infinity`, k256/src/arithmetic/affine.rs:45-49); NULL flags = none.
 *             The identity is encoded as x = y = 0 with flag 1 on output.
 *   projective inputs (batch_normalize only): n * 3L bytes X || Y || Z, canonical big-endian.
 *   ECGPU_BIGN256 (bign-curve256v1, `bignp256`) is the exception to "big-endian": its field elements and scalars travel
 *             LITTLE-endian, as in the reference (`FIELD_ENDIANNESS = LittleEndian`, bignp256/src/lib.rs:102); everything
 *             else about the records is the same.
 *
 * Ownership: the caller owns every buffer passed in; the library keeps no pointer after return.
 * Device memory, streams and the precomputed basepoint tables belong to the context
 * (the analogue of the reference's `static BASEPOINT_TABLE: LazyLock<..>`,
 * k256/src/arithmetic/tables.rs:18, primeorder/src/tables/basepoint.rs:29-31).
 *
 * Errors: every function returns ECGPU_OK (0) or a negative code; nothing aborts or throws across
 * the ABI.  Decoding failures mirror the reference's `CtOption`/`Error` results; the arithmetic
 * itself is total (complete formulas).  There is NO CPU fallback: without a usable gfx950 device
 * ecgpu_init fails with ECGPU_ERR_NO_DEVICE.
 *
 * Secret scalars: every entry point whose name does not end in `_ct` / `_ct_dev` is VARIABLE-TIME in its scalars (zero
 * digits are skipped; comb-table entries, per-point table entries and Pippenger buckets are addressed by scalar bits; the
 * kernels' duration and memory access pattern depend on the scalar).  Those stand behind the reference's `*_vartime` names —
 * `MulVartime::mul_vartime` (primeorder/src/projective.rs:888-921), `LinearCombination::lincomb_vartime` (:498-510,
 * k256/src/arithmetic/mul.rs:100-108), `MulByGeneratorVartime::{mul_by_generator_vartime,
 * mul_by_generator_and_mul_add_vartime}` (:923-940, k256 mul.rs:205-232,296-310) — and is meant for PUBLIC scalars:
 * signature verification, MSMs over public data, batch derivation of public values.  It is NOT a replacement for the
 * constant-time `Mul` / `mul_by_generator` / `lincomb` (primeorder/src/projective.rs:532-557, tables/lookup.rs:43-65)
 * when the scalar is a long-term secret and an attacker can observe the device (timing of a shared GPU, its memory
 * traffic).  ecgpu_batch_mul_base* and ecgpu_batch_ecdh accept whatever scalars they are given: a caller that passes
 * private keys there has decided that its threat model allows it; the results are the same group elements either way.
 * For secret scalars there are the uniform-schedule entry points ecgpu_batch_mul_base_ct, ecgpu_batch_mul_ct,
 * ecgpu_batch_ecdh_ct and ecgpu_lincomb_ct (below): the reference's constant-time drivers — fixed digit count, every table entry read and one
 * kept under a mask, complete additions — at 1.2-7x the cost of the variable-time kernels.
 *
 * Threading: a context may be used from one thread at a time (calls serialise on its stream);
 * create one context per thread / per GPU for concurrency.  The host-pointer forms of the per-unit batch calls
 * (everything except ecgpu_point_sum and ecgpu_batch_normalize) run batches of 2^19 units and more as a
 * pipeline over chunks of 2^18 units (ecgpu_msm: from 2^23 terms, as partial MSMs over chunks of 2^22 terms): two helper threads, alive for the duration of the call, move the next chunk in and
 * the previous one out on their own streams.  A pipelined call is not all-or-nothing: when a chunk fails validation
 * (ECGPU_ERR_SCALAR_RANGE / ECGPU_ERR_POINT) the results of the chunks before it may already have been written to the
 * caller's output buffers; below 2^19 units nothing is written on error.
 */
#ifndef ECGPU_H
#define ECGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ecgpu_ctx ecgpu_ctx;

enum { ECGPU_K256 = 0, ECGPU_P256 = 1, ECGPU_P384 = 2, ECGPU_SM2 = 3, ECGPU_P224 = 4, ECGPU_P192 = 5, ECGPU_P521 = 6, ECGPU_BP256 = 7, ECGPU_BP384 = 8, ECGPU_BP256T1 = 9, ECGPU_BP384T1 = 10, ECGPU_BIGN256 = 11 };

enum {
    ECGPU_OK = 0,
    ECGPU_ERR_CURVE = -1,        /* unknown curve id */
    ECGPU_ERR_SCALAR_RANGE = -2, /* a scalar >= n          (Scalar::from_repr -> None) */
    ECGPU_ERR_POINT = -3,        /* coordinate >= p or not on curve (AffinePoint::from_coordinates
                                    -> None, primeorder/src/affine.rs:100-109) */
    ECGPU_ERR_NO_DEVICE = -4,    /* no gfx950 device / HIP runtime unusable */
    ECGPU_ERR_HIP = -5,          /* a HIP call failed; see ecgpu_last_error */
    ECGPU_ERR_OOM = -6,          /* device or host allocation failed */
    ECGPU_ERR_ARG = -7           /* NULL / inconsistent argument */
};

/* ---- context -------------------------------------------------------------------------------- */

/* Number of gfx950 devices the HIP runtime shows this process (0: none / runtime unusable) — what a single-process
 * caller sizes its ecgpu_group from (elliptic-curves_amd/rust/ecgpu_shim.rs `NODE`). */
int ecgpu_device_count(void);

/* Creates a context on HIP device `device` (as numbered by the HIP runtime in this process).
 * Basepoint tables are built lazily on first use per curve and kept on the device: ONE table per (device, curve, width)
 * for the whole process, shared by every context on that device and freed with the last of them — the analogue of the
 * reference's process-wide `LazyLock<BasepointTable>` (k256/src/arithmetic/tables.rs:18).  Which width a call gets is the table
 * policy's decision (below: "the generator (comb) tables and their footprint").  If a table does not fit (k256's widest
 * is 21.5 GB) its comb width is lowered two bits at a time (a quarter of the memory, one or two more additions per
 * scalar, identical results) down to 16 bits before ECGPU_ERR_OOM is returned. */
int ecgpu_init(ecgpu_ctx **ctx, int device);
void ecgpu_destroy(ecgpu_ctx *ctx);

/* Human-readable text of the last error on this context (valid until the next call). */
const char *ecgpu_last_error(const ecgpu_ctx *ctx);

/* Field-byte length L of a curve (32 / 48 / 28 / 24 / 66), 0 for a bad id. */
size_t ecgpu_field_bytes(int curve);

/* Use an externally owned HIP stream (e.g. torch's current stream) for all later work on this
 * context; NULL restores the context's own stream.  `stream` is a hipStream_t. */
int ecgpu_set_stream(ecgpu_ctx *ctx, void *stream);

/* Page-locked host memory for the buffers of the host-pointer entry points.  Any host pointer works there, but
 * transfers from ordinary (pageable) memory go through the driver's bounce buffers at ~8 GB/s, which is 20x the
 * compute time of a fixed-base batch; from memory allocated here they run at PCIe DMA speed (DESIGN.md section 7 has
 * both numbers).  Returns NULL on failure.  ecgpu_host_free(NULL) is a no-op. */
void *ecgpu_host_alloc(ecgpu_ctx *ctx, size_t bytes);
void ecgpu_host_free(ecgpu_ctx *ctx, void *p);

/* Device memory for the *_dev entry points, for callers that have no HIP binding of their own (a Rust or C host that
 * wants its inputs to stay resident between calls; torch callers pass tensor data_ptr()s instead).  Allocations are
 * 256-byte aligned and live on the context's device until ecgpu_dev_free (ecgpu_destroy does not release them: free
 * them first).  The two copies are synchronous (they return when the bytes have arrived) and ordered after earlier
 * work on the context stream. */
void *ecgpu_dev_alloc(ecgpu_ctx *ctx, size_t bytes);
void ecgpu_dev_free(ecgpu_ctx *ctx, void *d_ptr);
int ecgpu_copy_to_device(ecgpu_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int ecgpu_copy_to_host(ecgpu_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- the generator (comb) tables and their footprint ----------------------------------------------------------------------
 * `mul_by_generator` reads a precomputed table of multiples of G that lives in device memory, shared by all contexts of a device:
 * ceil((bits - 1) / w) windows of 2^(w-1) affine entries (2L bytes each), w = the window width.  Wider tables mean fewer additions
 * per scalar and more memory and build time — k256: 16 bits = 34 MB built in ~1 ms, 15 additions per scalar, 0.83 ms per 2^20
 * scalars; 22 bits = 1.6 GB in ~6 ms, 11 additions, 0.73 ms; 26 bits = 21.5 GB in ~55 ms, 9 additions, 0.61 ms (measured: profiles/r05/table_tiers.txt, crossover batch sizes in
 * INTEGRATION.md).  The reference makes the same trade with its WINDOW_SIZE constants and builds lazily
 * (k256/src/arithmetic/mul.rs:191-192 prices a table half the size at 3 %; primeorder/src/tables/basepoint.rs:29-76).
 *
 * Default (ECGPU_TABLE_ADAPTIVE): the width follows the number of generator multiplications the DEVICE has been asked for so far,
 * the call being served included: 16 bits below 2^26, 22 bits below 2^29, then the curve's widest (k256 26, p256 / sm2 / the
 * other 256- and 224- / 192-bit sets 24, p384 / p521 / bp384 20) — each step where the time lost to the narrower table so far equals
 * the build time of the next one, so a caller never pays more than twice the best fixed choice and a 1,024-scalar call never
 * allocates gigabytes.  A step up costs the calling thread the build once — the call that crosses a tier waits for its stream and
 * builds the next table before it returns (k256: 1 / 6 / 55 ms), on an asynchronous context too —; the narrower table is freed when
 * its last user lets go.  Until a tier is reached the rate is the narrower table's (k256, per 2^20 scalars: ~0.93 ms at 16 bits,
 * ~0.74 at 22, ~0.59 at 26; bench.py prints the three as `fixed_k256_tier_ms`, its headline uses ECGPU_TABLE_EAGER).
 * ECGPU_TABLE_EAGER: the widest table at the first call (a long-lived service: pay ~55 ms once at start-up).
 * A table that does not fit is replaced by one two bits narrower, down to 16 bits, before ECGPU_ERR_OOM is returned. */
enum { ECGPU_TABLE_ADAPTIVE = 0, ECGPU_TABLE_EAGER = 1 };
int ecgpu_set_table_policy(ecgpu_ctx *ctx, int policy);

/* No comb table built for this context from now on takes more than max_table_bytes of device memory (0 = no limit, the default): the
 * width is lowered until the table fits the budget.  Tables already in use are kept until the next call on their curve. */
int ecgpu_set_table_budget(ecgpu_ctx *ctx, size_t max_table_bytes);

/* The comb table `curve` uses on this context right now: its window width (0 = none built yet), its bytes of device memory and
 * the wall time its construction took (whichever context of the device paid it).  Any of the three pointers may be NULL. */
int ecgpu_base_table_info(ecgpu_ctx *ctx, int curve, int *window_bits, size_t *table_bytes, double *build_ms);

/* Pins the width for later generator multiplications on `curve` to exactly window_bits (4..26), whatever the policy and the
 * budget; 0 returns the curve to the policy.  The table is rebuilt on next use.  A tuning knob, like the reference's WINDOW_SIZE
 * constants (k256/src/arithmetic/tables.rs:12, p384/src/arithmetic/tables.rs:8). */
int ecgpu_set_base_window(ecgpu_ctx *ctx, int curve, int window_bits);

/* Pippenger window width c for later ecgpu_msm* calls (4..16), 0 = choose from n (default). */
int ecgpu_set_msm_window(ecgpu_ctx *ctx, int window_bits);

/* Asynchronous mode for the device-pointer (`_dev`) entry points.  on != 0: a `_dev` call returns ECGPU_OK as soon as its
 * kernels are queued on the context's stream — argument errors are still reported at once — and the input checks
 * (ECGPU_ERR_SCALAR_RANGE, ECGPU_ERR_POINT) of all calls since the last ecgpu_synchronize are reported by the next
 * ecgpu_synchronize (outputs of the offending call are then unspecified, as in the synchronous mode).  Back-to-back batches
 * keep the GPU busy instead of paying a host round trip per call (0.70 -> 0.67 ms per 2^20-scalar fixed-base batch).
 * Buffers handed to a queued call must stay valid until ecgpu_synchronize.  ecgpu_last_timing waits for the stream.
 * Host-pointer entry points stay synchronous: on an asynchronous context they first wait for the queued work, whose
 * errors stay deferred.  ecgpu_set_async itself synchronises and returns what ecgpu_synchronize would.
 * ecgpu_synchronize on a synchronous context just waits for the stream. */
int ecgpu_set_async(ecgpu_ctx *ctx, int on);
int ecgpu_synchronize(ecgpu_ctx *ctx);

/* MSM lanes: with lanes = 2 .. 4 on an asynchronous context, consecutive ecgpu_msm_dev calls rotate over that many internal
 * streams, each with a workspace of its own, so that several independent MSMs are in flight: the sort and the reduction tail of
 * one (bandwidth- and latency-bound) run beside the accumulation of the other (issue-bound) — 8-12 % more MSMs per second on
 * one GPU (profiles/r03/msm_lanes.txt); the time of a single MSM does not change.  A lane starts after the work queued on
 * the context's stream at the time of the call (its inputs).  Ordering rules while MSMs are in flight on the lanes:
 *   - every MSM in flight needs output buffers of its own, and its INPUT buffers must stay untouched until it has finished;
 *   - any other entry point called later on this context (a batch call, ecgpu_point_sum_dev, ecgpu_copy_to_host, ...) first
 *     waits for all lanes, so it may read an MSM's output or overwrite its inputs;
 *   - a further ecgpu_msm_dev does NOT wait (that is the point), and neither does work the caller enqueues itself on the
 *     context's stream (ecgpu_set_stream): for those, inputs and outputs belong to the lanes until ecgpu_synchronize.
 * lanes = 1 (the default) restores one stream.  Synchronous contexts are unaffected.  Returns ECGPU_ERR_ARG for other values. */
int ecgpu_set_msm_lanes(ecgpu_ctx *ctx, int lanes);

/* Zeroes every staging and scratch buffer the context owns on the device (inputs copied in by host-pointer calls, projective
 * results, batch-inversion products, signature scratch, MSM workspaces) and waits for it.  The uniform-schedule (`_ct`) entry
 * points do this by themselves for what they touch — the reference keeps secret scalars and shared secrets in
 * zeroize-on-drop types (`NonZeroScalar`, `SharedSecret`) —; a caller that has passed secrets through a variable-time name
 * (private keys to ecgpu_batch_mul_base) calls this before it lets go of the context.  Buffers the CALLER allocated
 * (ecgpu_dev_alloc, ecgpu_host_alloc, torch tensors) are the caller's to wipe.  Basepoint tables are public data and stay. */
int ecgpu_wipe(ecgpu_ctx *ctx);

/* ---- host-pointer entry points (copy in, compute on the GPU, copy out) ------------------------ */

/* out[i] = k[i] * G.
 * Batch form of `ProjectivePoint::mul_by_generator` (k256/src/arithmetic/mul.rs:180-203) and
 * `MulBackend::mul_by_generator` (primeorder/src/mul_backend.rs:15-17,
 * p256/src/arithmetic/tables.rs:33-43), followed by `to_affine`. */
int ecgpu_batch_mul_base(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, size_t n,
                         uint8_t *out_xy, uint8_t *out_inf);

/* The same with SEC1-compressed output, split in two arrays: out_x[i] = x of k[i] * G (L bytes), out_tag[i] = 0x02 / 0x03
 * (y even / odd), 0x00 with x = 0 for the identity — tag || x is `to_sec1_point(true)` (primeorder/src/affine.rs:387-401,
 * k256/src/arithmetic/affine.rs:341), what `PublicKey::to_sec1_bytes` ships.  Half the bytes of the x || y form come
 * back over PCIe, which is what bounds the host-pointer rate. */
int ecgpu_batch_mul_base_compressed(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, size_t n, uint8_t *out_x,
                                    uint8_t *out_tag);

/* out[i] = k[i] * P[i].
 * Batch form of `impl Mul<Scalar> for ProjectivePoint` (k256/src/arithmetic/mul.rs:249-274,
 * primeorder/src/projective.rs:847-886) / `MulVartime` (:888-921), followed by `to_affine`. */
int ecgpu_batch_mul(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy,
                    const uint8_t *points_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);

/* out = sum_i k[i] * P[i].
 * `LinearCombination<[(ProjectivePoint, Scalar)]>::{lincomb, lincomb_vartime}`
 * (k256/src/arithmetic/mul.rs:84-109, primeorder/src/projective.rs:480-511); n == 0 gives the
 * identity. out_xy holds 2L bytes, out_inf one byte. */
int ecgpu_msm(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy,
              const uint8_t *points_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);

/* out[i] = a[i] * G + b[i] * P[i].
 * Batch form of `MulByGeneratorVartime::mul_by_generator_and_mul_add_vartime`
 * (primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310) — the ECDSA/Schnorr
 * verification shape. */
int ecgpu_batch_mul_base_and_mul_add(ecgpu_ctx *ctx, int curve, const uint8_t *a_scalars,
                                     const uint8_t *b_scalars, const uint8_t *points_xy,
                                     const uint8_t *points_inf, size_t n, uint8_t *out_xy,
                                     uint8_t *out_inf);

/* Projective -> affine for n points.
 * `BatchNormalize::batch_normalize` (k256/src/arithmetic/projective.rs:367-391,
 * primeorder/src/projective.rs:452-478). */
int ecgpu_batch_normalize(ecgpu_ctx *ctx, int curve, const uint8_t *points_xyz, size_t n,
                          uint8_t *out_xy, uint8_t *out_inf);

/* ---- device-pointer entry points ------------------------------------------------------------- *
 * Same semantics, all buffers already resident in this context's device memory (e.g. torch CUDA
 * tensors' data_ptr()).  Work is enqueued on the context stream; the call returns after the
 * status word has been read back (one small D2H copy + stream sync), so errors are reported
 * synchronously exactly like the host-pointer forms.  The context stream is a non-blocking stream of its own: inputs
 * written by work on another stream (torch's current stream, the legacy default stream) are NOT ordered before these
 * calls — either make that stream the context's with ecgpu_set_stream, or synchronise it first. */

int ecgpu_batch_mul_base_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, size_t n,
                             void *d_out_xy, void *d_out_inf);
int ecgpu_batch_mul_base_compressed_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, size_t n, void *d_out_x,
                                        void *d_out_tag);
int ecgpu_batch_mul_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy,
                        const void *d_points_inf, size_t n, void *d_out_xy, void *d_out_inf);
int ecgpu_msm_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy,
                  const void *d_points_inf, size_t n, void *d_out_xy, void *d_out_inf);
int ecgpu_batch_mul_base_and_mul_add_dev(ecgpu_ctx *ctx, int curve, const void *d_a_scalars,
                                         const void *d_b_scalars, const void *d_points_xy,
                                         const void *d_points_inf, size_t n, void *d_out_xy,
                                         void *d_out_inf);
int ecgpu_batch_normalize_dev(ecgpu_ctx *ctx, int curve, const void *d_points_xyz, size_t n,
                              void *d_out_xy, void *d_out_inf);

/* Sum of n affine points (out = P[0] + ... + P[n-1]) — the combine step of a sharded MSM:
 * each GPU's partial `lincomb` result is all-gathered and added here.
 * Uses the same `Add` as `impl Sum for ProjectivePoint` (k256 projective.rs, primeorder
 * projective.rs `Sum` impls). */
int ecgpu_point_sum(ecgpu_ctx *ctx, int curve, const uint8_t *points_xy, const uint8_t *points_inf,
                    size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_point_sum_dev(ecgpu_ctx *ctx, int curve, const void *d_points_xy,
                        const void *d_points_inf, size_t n, void *d_out_xy, void *d_out_inf);

/* ---- an MSM whose terms are spread over several GPUs ------------------------------------------------ *
 * sum_i k_i P_i = sum over GPUs of the sum over that GPU's terms, and the bucket method is linear up to its last step:
 * every GPU runs the pipeline on its own terms down to the per-window partial sums ("parts", ecgpu_msm_parts_bytes bytes of
 * opaque internal-form points — 41 KiB for k256 at c = 16), the parts are exchanged (one all-gather: RCCL over xGMI in a
 * torch.distributed job, peer copies inside ecgpu_group_msm), and ONE combining step — window sums over all GPUs, then the
 * chain of doublings over the windows — produces the result.  The serial tail of the method runs once, not once per GPU
 * plus a point sum.  `plan_terms` is the term count the window width is chosen from and must be the same on every GPU
 * (at least the largest shard); so must ecgpu_set_msm_window.  `lincomb` semantics as for ecgpu_msm_dev. */
size_t ecgpu_msm_parts_bytes(ecgpu_ctx *ctx, int curve, size_t plan_terms);
/* The Pippenger window width (bits) the two halves use for `plan_terms` on this context — ecgpu_set_msm_window's override
 * or the measured default; every GPU of a sharded MSM must report the same value (ecgpu_group_msm_dev checks it).
 * 0 for an unknown curve id. */
int ecgpu_msm_plan_window(ecgpu_ctx *ctx, int curve, size_t plan_terms);
int ecgpu_msm_parts_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy, const void *d_points_inf,
                        size_t n, size_t plan_terms, void *d_parts);
/* Throughput form for back-to-back sharded MSMs: on an asynchronous context (ecgpu_set_async) with ecgpu_set_msm_lanes(L > 1),
 * ecgpu_msm_parts_dev runs on one of L rotating internal streams + workspaces, ordered after what the context's stream holds at
 * the call (the inputs) — so the exchange and the combining half of MSM i - 1, which the caller keeps on the context's stream,
 * run beside the accumulation of MSM i:
 *     ecgpu_msm_parts_dev(i, d_parts[i % L]);
 *     ecgpu_msm_parts_join_dev(d_parts[(i - 1) % L]);  all-gather(i - 1);  ecgpu_msm_finish_dev(i - 1);
 * ecgpu_msm_parts_join_dev makes the context's stream wait (on the device; the host does not) for the local half that wrote
 * `d_parts`; until then the record belongs to its lane.  A no-op for a record written on the context's own stream.  A lane keeps
 * track of ONE record: a local half that is still un-joined when its lane comes around again (more halves in flight than lanes) is
 * joined by that call — correct, but the context's stream then waits for it.
 * ecgpu_msm_finish_dev does not wait for local halves in flight.  One MSM alone gains nothing from lanes. */
int ecgpu_msm_parts_join_dev(ecgpu_ctx *ctx, const void *d_parts);
/* d_parts_all: nranks consecutive parts records (the all-gather's output). */
int ecgpu_msm_finish_dev(ecgpu_ctx *ctx, int curve, const void *d_parts_all, int nranks, size_t plan_terms, void *d_out_xy,
                         void *d_out_inf);

/* ---- all GPUs of a node from ONE process (`ecgpu_init(ctx**, devices, ndev)`; 8e) --------------------- *
 * A group owns one context per listed device and runs one worker thread per device for the duration of a call: what a
 * single-process caller — the Rust `lincomb` of the reference, which knows nothing about ranks — uses to reach the 8 GPUs
 * of a node.  (A torch.distributed job with one process per GPU uses ecgpu_msm_parts_dev / ecgpu_msm_finish_dev and its own
 * collective instead: bench.py.)  A device may be listed more than once (independent contexts; how the exchange is tested
 * on a one-GPU box).  Batch calls cut the index range into one contiguous slice per GPU and need no exchange; the MSM
 * cuts the terms the same way, runs ecgpu_msm_parts_dev per GPU and has ONE exchange step: RCCL ncclAllGather of the
 * per-window partial sums over xGMI when librccl can be dlopen()ed and the devices are distinct, a peer copy into
 * GPU 0 otherwise or after ecgpu_group_set_exchange(ECGPU_EXCHANGE_PEER); then ecgpu_msm_finish_dev once.
 * Results are identical to the single-GPU calls.  Not thread-safe: one call at a time per group. */
typedef struct ecgpu_group ecgpu_group;
int ecgpu_group_init(ecgpu_group **group, const int *devices, int ndev);
void ecgpu_group_destroy(ecgpu_group *group);
int ecgpu_group_size(const ecgpu_group *group);
ecgpu_ctx *ecgpu_group_ctx(ecgpu_group *group, int i);              /* borrowed: member i's context */
const char *ecgpu_group_last_error(const ecgpu_group *group);
const char *ecgpu_group_exchange(const ecgpu_group *group);         /* "rccl" or "peer" */
/* why: "rccl: ncclCommInitAll over 8 devices", "peer: ecgpu_group_set_exchange(ECGPU_EXCHANGE_PEER)", "peer: duplicate devices in the group",
 * "peer: librccl could not be loaded (...)", "peer: ncclCommInitAll failed (...)", "peer: ncclAllGather failed (...)": a
 * group whose RCCL exchange fails at run time falls back to peer copies for that call and all later ones */
const char *ecgpu_group_exchange_reason(const ecgpu_group *group);
/* Choose the exchange of an initialised group.  ECGPU_EXCHANGE_PEER: peer copies from now on (the communicators are released);
 * ECGPU_EXCHANGE_RCCL: ECGPU_OK if the group exchanges over RCCL, ECGPU_ERR_HIP if it does not (a caller who wants "RCCL or
 * nothing" instead of the silent fallback; the reason is in ecgpu_group_last_error).  The library reads no environment variable. */
enum { ECGPU_EXCHANGE_PEER = 1, ECGPU_EXCHANGE_RCCL = 2 };
int ecgpu_group_set_exchange(ecgpu_group *group, int mode);
int ecgpu_group_set_msm_window(ecgpu_group *group, int window_bits);
/* The exchange step of ecgpu_group_msm* never waits longer than this (default 10 s; seconds > 0): every member polls its exchange
 * stream against the deadline instead of blocking on it.  A collective that fails on any member, or has not completed by then
 * (the way RCCL has failed on this hardware is a hang, not an error code), is given up — communicators aborted (ncclCommAbort),
 * fresh exchange streams — and the call completes over peer copies, as do all later calls; ecgpu_group_exchange_reason reports it.
 * ECGPU_ERR_HIP only if the peer copies miss the deadline too. */
int ecgpu_group_set_exchange_timeout(ecgpu_group *group, double seconds);
/* `lincomb` over all GPUs of the group, host buffers (every GPU uploads its own shard over its own PCIe link). */
int ecgpu_group_msm(ecgpu_group *group, int curve, const uint8_t *scalars, const uint8_t *points_xy,
                    const uint8_t *points_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);
/* The same with the shards already resident: d_scalars[i] / d_points_xy[i] / d_points_inf[i] (the array or any entry may be
 * NULL) live on member i's device and hold n_per_device[i] terms; out_xy / out_inf are host buffers. */
int ecgpu_group_msm_dev(ecgpu_group *group, int curve, const void *const *d_scalars, const void *const *d_points_xy,
                        const void *const *d_points_inf, const size_t *n_per_device, uint8_t *out_xy, uint8_t *out_inf);
/* ecgpu_batch_mul_base / ecgpu_batch_mul over all GPUs of the group (index-range slices, no exchange). */
int ecgpu_group_batch_mul_base(ecgpu_group *group, int curve, const uint8_t *scalars, size_t n, uint8_t *out_xy,
                               uint8_t *out_inf);
int ecgpu_group_batch_mul(ecgpu_group *group, int curve, const uint8_t *scalars, const uint8_t *points_xy,
                          const uint8_t *points_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);

/* ecgpu_ecdsa_verify_batch / ecgpu_ecdsa_verify_msg_batch / ecgpu_ecdsa_recover_batch over all GPUs of the group (index-range
 * slices, no exchange; host buffers). */
int ecgpu_group_ecdsa_verify_batch(ecgpu_group *group, int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s,
                                   const uint8_t *q_xy, size_t n, int reject_high_s, uint8_t *ok);
int ecgpu_group_ecdsa_verify_msg_batch(ecgpu_group *group, int curve, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len,
                                       const uint8_t *sigs, size_t n, int reject_high_s, uint8_t *ok);
int ecgpu_group_ecdsa_recover_batch(ecgpu_group *group, int curve, const uint8_t *z, const uint8_t *r, const uint8_t *s,
                                    const uint8_t *recid, size_t n, int reject_high_s, uint8_t *out_xy, uint8_t *ok);

/* ---- introspection / measurement ---------------------------------------------------------------- */

/* k256 GLV split on the device: k -> (r1, r2) with r1 + r2*lambda = k (mod n), canonical scalars
 * before sign folding — `glv::decompose_scalar` (k256/src/arithmetic/mul/glv.rs:149-156).
 * Host buffers of n*32 bytes each. */
int ecgpu_k256_glv_decompose(ecgpu_ctx *ctx, const uint8_t *scalars, size_t n, uint8_t *r1,
                             uint8_t *r2);

/* Batch ECDSA verification — SURVEY.md §8(f) rank 1, the dominant caller of
 * `mul_by_generator_and_mul_add_vartime` (primeorder/src/mul_backend.rs:29-40, k256/src/arithmetic/mul.rs:303-310).
 * The equation is `ecdsa::hazmat::verify_prehashed` (ecdsa 0.17.0, un-vendored, Cargo.lock:428-429; instantiated
 * at p256/src/ecdsa.rs:69, p384/src/ecdsa.rs, k256/src/ecdsa.rs:104-106), i.e. SEC1 v2 §4.1.4.  Per element i:
 *   z  L bytes big-endian: the leftmost L bytes of the message digest as an integer (`bits2field`); it is
 *      reduced mod n on the device like `Scalar::reduce`
 *   r, s  L bytes big-endian each; q_xy  the public key, affine x||y
 *   ok[i] = 1 iff 1 <= r, s < n, (reject_high_s == 0 or s <= (n-1)/2 — pass the curve's `NORMALIZE_S`,
 *      true for k256), Q is a valid non-identity curve point, R = (z/s) G + (r/s) Q is not the identity and
 *      x(R) mod n == r;  otherwise 0.  A bad element never fails the batch: the return value reports only
 *      argument / device errors. */
int ecgpu_ecdsa_verify_batch(ecgpu_ctx *ctx, int curve, const uint8_t *z, const uint8_t *r,
                             const uint8_t *s, const uint8_t *q_xy, size_t n, int reject_high_s,
                             uint8_t *ok);
int ecgpu_ecdsa_verify_batch_dev(ecgpu_ctx *ctx, int curve, const void *d_z, const void *d_r,
                                 const void *d_s, const void *d_q_xy, size_t n, int reject_high_s,
                                 void *d_ok);

/* Batch ECDSA verification of messages — `signature::Verifier::verify(msg, &signature)` of `ecdsa::VerifyingKey<C>`: the
 * curve's digest (the reference's `DigestAlgorithm`: SHA-256 for k256 / p256 / brainpoolP256 — k256/src/ecdsa.rs:117-119,
 * p256/src/ecdsa.rs:72-74 —, SHA-384 for p384 / brainpoolP384, SHA-224 for p224, SHA-512 for p521) is computed on the device,
 * z = bits2field(digest) (the leftmost L bytes, left-padded when the digest is shorter), then ecgpu_ecdsa_verify_batch.
 * The reference's message-level vectors: its Wycheproof blobs (k256/src/ecdsa.rs:263-384 and the new_wycheproof_test! calls).
 *   q_xy n*2L bytes, msgs n*msg_len bytes (one uniform length per call, 0 allowed), sigs n*2L bytes (r || s, the fixed-size
 *   `Signature::from_slice` form; DER is parsed by the caller).  ECGPU_ERR_CURVE for p192 (no DigestAlgorithm), sm2, bign256. */
int ecgpu_ecdsa_verify_msg_batch(ecgpu_ctx *ctx, int curve, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len,
                                 const uint8_t *sigs, size_t n, int reject_high_s, uint8_t *ok);
int ecgpu_ecdsa_verify_msg_batch_dev(ecgpu_ctx *ctx, int curve, const void *d_q_xy, const void *d_msgs, size_t msg_len,
                                     const void *d_sigs, size_t n, int reject_high_s, void *d_ok);

/* Batch ECDSA public-key recovery — `VerifyingKey::recover_from_prehash(prehash, &signature, recovery_id)` of the `ecdsa`
 * crate (0.17.0, un-vendored, Cargo.lock:428-429), which k256 / p256 re-export and the reference tests with its own vectors
 * (k256/src/ecdsa.rs:170-262: RECOVERY_TEST_VECTORS and the Ethereum example; p256/tests/ecdsa.rs:20-25).  It is a caller
 * of `ProjectivePoint::lincomb(&[(G, u1), (R, u2)])`.  Per element i (SEC1 v2 4.1.6 for ONE candidate):
 *   z, r, s  as for ecgpu_ecdsa_verify_batch
 *   recid    one byte, `RecoveryId::to_byte`: bit 0 = y(R) is odd, bit 1 = x(R) = r + n ("x reduced"); > 3 does not parse
 *   R = decompress(r or r + n, bit 0);  key = -(z/r) G + (s/r) R
 *   ok[i] = 1 and out_xy[i] = the key (affine x||y) iff 1 <= r, s < n, (reject_high_s == 0 or s <= (n-1)/2: the
 *      `verify_prehash` the crate runs on the recovered key applies the curve's NORMALIZE_S — pass 1 for k256), recid <= 3,
 *      the candidate x is below p and on the curve, and the key is not the identity;  otherwise ok[i] = 0 and a zero record.
 * ECGPU_ERR_CURVE for sm2 / bign256 (not ECDSA). */
int ecgpu_ecdsa_recover_batch(ecgpu_ctx *ctx, int curve, const uint8_t *z, const uint8_t *r,
                              const uint8_t *s, const uint8_t *recid, size_t n, int reject_high_s,
                              uint8_t *out_xy, uint8_t *ok);
int ecgpu_ecdsa_recover_batch_dev(ecgpu_ctx *ctx, int curve, const void *d_z, const void *d_r,
                                  const void *d_s, const void *d_recid, size_t n, int reject_high_s,
                                  void *d_out_xy, void *d_ok);

/* Batch BIP340 Schnorr verification over secp256k1 — `VerifyingKey::verify_raw`
 * (k256/src/schnorr/verifying.rs:76-99) without the hash: per element
 *   e  32 bytes big-endian = tagged_hash("BIP0340/challenge", r || pk || m), computed by the caller; reduced
 *      mod n on the device like `<Scalar as Reduce<FieldBytes>>::reduce`
 *   r, s  the two halves of the signature (k256/src/schnorr.rs:132-150: r < p, 0 < s < n)
 *   p_xy  the verifying key's affine point, i.e. the even-y lift of the 32-byte public key
 *      (`VerifyingKey::from_bytes`; use ecgpu_batch_decompress with y_is_odd = 0)
 *   ok[i] = 1 iff the fields are in range, P is on the curve, R = s G - e P is not the identity, y(R) is even
 *      and x(R) == r. */
int ecgpu_schnorr_verify_batch(ecgpu_ctx *ctx, const uint8_t *e, const uint8_t *r, const uint8_t *s,
                               const uint8_t *p_xy, size_t n, uint8_t *ok);
int ecgpu_schnorr_verify_batch_dev(ecgpu_ctx *ctx, const void *d_e, const void *d_r, const void *d_s,
                                   const void *d_p_xy, size_t n, void *d_ok);

/* Batch SM2DSA verification on the prehash — `PrehashVerifier::verify_prehash` of sm2::dsa::VerifyingKey
 * (sm2/src/dsa/verifying.rs:138-171; GB/T 32918.2, draft-shen-sm2-ecdsa 5.3), curve sm2 only.  Per element
 *   e     32 bytes big-endian = SM3(ZA || M), computed by the caller (ZA hashes the signer's identity and key,
 *         sm2/src/dsa.rs); reduced mod n on the device like `Scalar::reduce`
 *   r, s  the signature halves, q_xy the public key's affine point
 *   ok[i] = 1 iff 1 <= r, s < n, t = r + s mod n != 0, Q is a valid non-identity curve point and
 *         r == e + x(s G + t Q) mod n   (the `lincomb` at verifying.rs:161). */
int ecgpu_sm2dsa_verify_batch(ecgpu_ctx *ctx, const uint8_t *e, const uint8_t *r, const uint8_t *s, const uint8_t *q_xy,
                              size_t n, uint8_t *ok);
int ecgpu_sm2dsa_verify_batch_dev(ecgpu_ctx *ctx, const void *d_e, const void *d_r, const void *d_s, const void *d_q_xy,
                                  size_t n, void *d_ok);

/* SM2DSA verification of messages — `sm2::dsa::VerifyingKey::new(distid, public_key)?.verify(msg, &signature)`: the identity
 * hash Z = SM3(ENTL || ID || a || b || xG || yG || xA || yA) (`hash_z`, sm2/src/distid.rs:21-44) and e = SM3(Z || M)
 * (`hash_msg`, sm2/src/dsa/verifying.rs:126-130) are computed on the device, then the verification above.  The reference's
 * message-level vector: sm2/tests/sm2dsa.rs:16-35.
 *   distid  the signers' distinguishing identifier, distid_len <= 8191 bytes (ENTL is a 16-bit bit count), one per call
 *   q_xy    n*64 bytes, msgs n*msg_len bytes (one uniform length per call, 0 allowed), sigs n*64 bytes (r || s)
 *   ok[i] as for ecgpu_sm2dsa_verify_batch. */
int ecgpu_sm2dsa_verify_msg_batch(ecgpu_ctx *ctx, const uint8_t *distid, size_t distid_len, const uint8_t *q_xy,
                                  const uint8_t *msgs, size_t msg_len, const uint8_t *sigs, size_t n, uint8_t *ok);
int ecgpu_sm2dsa_verify_msg_batch_dev(ecgpu_ctx *ctx, const void *d_distid, size_t distid_len, const void *d_q_xy,
                                      const void *d_msgs, size_t msg_len, const void *d_sigs, size_t n, void *d_ok);

/* Batch bign verification on the prehash — `PrehashVerifier::verify_prehash` of bignp256::ecdsa::VerifyingKey
 * (bignp256/src/ecdsa/verifying.rs:100-155; STB 34.101.45-2013 §7.2), curve bign256 only.  Everything is little-endian, as
 * the crate's `FIELD_ENDIANNESS` (bignp256/src/lib.rs:90).  Per element
 *   h     32 bytes = belt-hash(message), as `verify_prehash` takes it (reduced mod q like `Scalar::reduce`)
 *   sigs  48 bytes S0 (16) || S1 (32) — `Signature::from_bytes`, bignp256/src/ecdsa.rs:72-88
 *   q_xy  the public key's affine point, 64 bytes
 *   ok[i] = 1 iff S0 != 0, 0 < S1 < q, Q is a valid non-identity curve point, R = ((S1 + H) mod q) G + (S0 + 2^128) Q is
 *         finite (the `lincomb` at verifying.rs:119-122) and S0 equals the first 16 bytes of
 *         belt-hash(OID(belt-hash) || x(R) as 32 bytes || h), hashed on the device (csrc/ecgpu_belt.h).
 * The reference's vector: bignp256/tests/ecdsa.rs:21-46. */
int ecgpu_bign_verify_batch(ecgpu_ctx *ctx, const uint8_t *h, const uint8_t *sigs, const uint8_t *q_xy, size_t n,
                            uint8_t *ok);
int ecgpu_bign_verify_batch_dev(ecgpu_ctx *ctx, const void *d_h, const void *d_sigs, const void *d_q_xy, size_t n,
                                void *d_ok);

/* bign verification of messages — `VerifyingKey::from_bytes(pk)?.verify(msg, &signature)` (bignp256/src/ecdsa/verifying.rs:
 * 157-169): h = belt-hash(msg) (`hash_msg`, :87-91) on the device, then the verification above.
 *   q_xy n*64 bytes, msgs n*msg_len bytes (one uniform length per call, 0 allowed), sigs n*48 bytes. */
int ecgpu_bign_verify_msg_batch(ecgpu_ctx *ctx, const uint8_t *q_xy, const uint8_t *msgs, size_t msg_len,
                                const uint8_t *sigs, size_t n, uint8_t *ok);
int ecgpu_bign_verify_msg_batch_dev(ecgpu_ctx *ctx, const void *d_q_xy, const void *d_msgs, size_t msg_len,
                                    const void *d_sigs, size_t n, void *d_ok);

/* The same verification from wire bytes — `VerifyingKey::from_bytes(pk)?.verify_raw(msg, sig)`
 * (k256/src/schnorr/verifying.rs:76-99,149-160): pk_x n*32 bytes (x-only keys, lifted on the device with even y),
 * msgs n*msg_len bytes (one uniform length per call, 0 allowed), sigs n*64 bytes (r || s).  The challenge
 * e = tagged_hash("BIP0340/challenge", r || pk || msg) is computed on the device (SHA-256).  ok[i] = 0 for keys that do
 * not lift, out-of-range signature halves, and signatures that do not verify. */
int ecgpu_schnorr_verify_raw_batch(ecgpu_ctx *ctx, const uint8_t *pk_x, const uint8_t *msgs, size_t msg_len,
                                   const uint8_t *sigs, size_t n, uint8_t *ok);
int ecgpu_schnorr_verify_raw_batch_dev(ecgpu_ctx *ctx, const void *d_pk_x, const void *d_msgs,
                                       size_t msg_len, const void *d_sigs, size_t n, void *d_ok);

/* Batch ECDH — `elliptic_curve::ecdh::diffie_hellman(secret, public)` (elliptic-curve 0.14.1, un-vendored; the
 * curves re-export it: k256/src/ecdh.rs, p256/src/ecdh.rs, p384/src/ecdh.rs): out_x[i] = the x-coordinate of
 * k_i * P_i as L big-endian bytes (`SharedSecret::raw_secret_bytes`), ok[i] = 1 unless the product is the identity
 * (k_i = 0).  Scalars >= n and points off the curve fail the call like ecgpu_batch_mul. */
int ecgpu_batch_ecdh(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy,
                     size_t n, uint8_t *out_x, uint8_t *ok);
int ecgpu_batch_ecdh_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy,
                         size_t n, void *d_out_x, void *d_ok);

/* ---- uniform-schedule ("constant-time shaped") variants for SECRET scalars --------------------------------------------
 * Same arguments, results and error behaviour as ecgpu_batch_mul_base / ecgpu_batch_mul / ecgpu_batch_ecdh, computed by the
 * reference's constant-time algorithms (elliptic-curves_amd/csrc/ecgpu_ctmul.h):
 *   ecgpu_batch_mul_base_ct   `ProjectivePoint::mul_by_generator` (k256/src/arithmetic/mul.rs:180-197) /
 *                             `BasepointTable::mul` (primeorder/src/tables/basepoint.rs:82-99), with 6-bit windows where the
 *                             reference has nibbles: one LUT of the 32 multiples e 2^(6 i) G per window, one signed digit
 *                             and one complete mixed addition per window (43 for a 256-bit scalar), no doublings; every
 *                             LUT is scanned in full (`LookupTable::select`, primeorder/src/tables/lookup.rs:43-65)
 *   ecgpu_batch_mul_ct        `impl Mul<Scalar> for ProjectivePoint` (primeorder/src/projective.rs:847-886 -> `lincomb`
 *                             :532-557 with one term; k256/src/arithmetic/mul.rs:112-163,249-274 on the two GLV halves):
 *                             table [P..8P] by complete additions (lookup.rs:30-38), 65 / 97 (k256: 2 x 33) digits, four
 *                             doublings (complete ones on k256; Jacobian ones, which have no exceptional case on a
 *                             prime-order curve, with the identity patched under a mask elsewhere) and one complete
 *                             addition per digit, no digit skipped, the accumulator starts at the identity
 *   ecgpu_batch_ecdh_ct       `diffie_hellman(secret, public)` (k256/src/ecdh.rs:56-60 over the `Mul` above): x of
 *                             ecgpu_batch_mul_ct
 *   ecgpu_lincomb_ct          `LinearCombination::lincomb` (primeorder/src/projective.rs:484-496 -> :532-557;
 *                             k256/src/arithmetic/mul.rs:84-98 -> :112-163): out = sum_i k_i P_i with one ecgpu_batch_mul_ct
 *                             multiplication per term and a tree of complete additions over the n products (256 per workgroup
 *                             and level).  The reference interleaves the terms on one accumulator; the group element is the
 *                             same, and the instruction and memory schedule here is a function of n alone.  n == 0 gives the
 *                             identity.  Cost: n uniform-schedule multiplications — for PUBLIC scalars ecgpu_msm (the bucket
 *                             method) is 50-100x faster from a few thousand terms on
 * What is guaranteed, and checked on the gfx950 ISA of the two kernels by tools/ct_isa_check.py (a register-level taint
 * analysis from every loaded record to every branch condition and every memory address; tests/test_ct_isa.py): no
 * conditional branch and no load / store address depends on the contents of a scalar or point record; range and
 * on-curve verdicts are computed for every element and reported through the usual error codes after the kernel.
 * What is not: the conversion to affine output that follows branches on "the result is the identity" (k = 0 or
 * P = identity), as `to_affine` must distinguish that case for the wire format; power and clock side channels of the
 * device are outside what an ISA-level argument can cover.  Measured cost: DESIGN.md §7. */
int ecgpu_batch_mul_base_ct(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_batch_mul_base_ct_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, size_t n, void *d_out_xy, void *d_out_inf);
int ecgpu_batch_mul_ct(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy, const uint8_t *points_inf,
                       size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_batch_mul_ct_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy, const void *d_points_inf,
                           size_t n, void *d_out_xy, void *d_out_inf);
int ecgpu_batch_ecdh_ct(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy, size_t n, uint8_t *out_x,
                        uint8_t *ok);
int ecgpu_batch_ecdh_ct_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy, size_t n, void *d_out_x,
                            void *d_ok);
int ecgpu_lincomb_ct(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_xy, const uint8_t *points_inf,
                     size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_lincomb_ct_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_xy, const void *d_points_inf,
                         size_t n, void *d_out_xy, void *d_out_inf);

/* Compressed points INTO the path (callers hold 33-byte SEC1 keys).  Like ecgpu_msm / ecgpu_batch_mul with
 * point i given as points_x[i] (L bytes, the curve's wire order) + points_tag[i]: 0x02 / 0x03 = the point with that x and even /
 * odd y (`FromSec1Point::from_sec1_point` of a compressed encoding -> `DecompressPoint::decompress`,
 * primeorder/src/affine.rs:183-200,352-366, k256/src/arithmetic/affine.rs:261-280), 0x00 (with x = 0) = the identity
 * (`Sec1Point::identity`).  The points are decoded on the device — one square root each, which costs about as much as 25 point
 * additions: at 2^24 k256 terms the decoding takes as long as the MSM itself — and the ordinary pipeline runs on the
 * result.  Any other tag, x >= p or an x that is on no point of the curve (the reference's `CtOption::None`) fails the call
 * with ECGPU_ERR_POINT.  Variable-time like the calls they feed; for secret scalars over compressed public keys decode with
 * ecgpu_batch_decompress_dev and pass the result to ecgpu_batch_mul_ct_dev. */
int ecgpu_msm_compressed(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_x, const uint8_t *points_tag,
                         size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_msm_compressed_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_x, const void *d_points_tag,
                             size_t n, void *d_out_xy, void *d_out_inf);
int ecgpu_batch_mul_compressed(ecgpu_ctx *ctx, int curve, const uint8_t *scalars, const uint8_t *points_x,
                               const uint8_t *points_tag, size_t n, uint8_t *out_xy, uint8_t *out_inf);
int ecgpu_batch_mul_compressed_dev(ecgpu_ctx *ctx, int curve, const void *d_scalars, const void *d_points_x,
                                   const void *d_points_tag, size_t n, void *d_out_xy, void *d_out_inf);

/* Batch point decompression — `DecompressPoint::decompress(x_bytes, y_is_odd)`
 * (primeorder/src/affine.rs:183-200, k256/src/arithmetic/affine.rs:261-280; SEC1 tag 0x02 / 0x03 = y_is_odd 0 / 1;
 * BIP340 `decompact` = y_is_odd 0).  xs n*L bytes big-endian, y_is_odd n bytes.  out_xy[i] = (x, y) with
 * y^2 = x^3 + a x + b and the requested parity and ok[i] = 1, or a zero record and ok[i] = 0 when x >= p or
 * no such y exists (the reference's `CtOption::None`).  Every parameter set, p224 included (p = 1 mod 4: the square root
 * is a fixed-schedule Tonelli-Shanks over the 2^96-element subgroup, ~20 times the work of the other curves' single
 * exponentiation; primefield/src/monty.rs:467-469). */
int ecgpu_batch_decompress(ecgpu_ctx *ctx, int curve, const uint8_t *xs, const uint8_t *y_is_odd,
                           size_t n, uint8_t *out_xy, uint8_t *ok);
int ecgpu_batch_decompress_dev(ecgpu_ctx *ctx, int curve, const void *d_xs, const void *d_y_is_odd,
                               size_t n, void *d_out_xy, void *d_ok);

/* Device-side known-answer tests of the arithmetic the kernels are built from — the same field / group code the CPU
 * host checks run (tests/hostcheck), here as gfx950 code, one lane per element; host buffers.
 * ecgpu_selftest_field: out[i] = op(a[i], b[i]) on canonical field elements (L bytes each; b may be NULL for unary ops).
 *   op 0 a + b, 1 a - b, 2 a * b, 3 a^2, 4 1/a by division steps (0 for 0), 5 -a, 7 2a, 8 pack / unpack round trip of the
 *   lazily reduced 2a + b, 9 the fused a*b - (a + b)*b, 10 1/a by Fermat, 11 sqrt(a) or 0, 12 a 25-step chain at the
 *   magnitudes the point formulas use, 13 / 14 the fused a*b - c and a^2 - c, 15 (k256) the reduction in assembly against the
 *   compiler's from the same columns (14 a b, or all ones on a mismatch), 16 (k256; n a multiple of 64) the row-parallel
 *   multiplication of the MSM's Horner chain: 13 a b of lane (i mod 4) of the wave in every lane i, 17 1/a by the variable-time
 *   division steps.  The reference's counterparts: `FieldElement::{add, sub, mul, square, invert,
 *   negate, double, sqrt}` (k256/src/arithmetic/field.rs, p256/src/arithmetic/field.rs, primefield/src/monty.rs).
 * ecgpu_selftest_point: out[i] = op(P[i], Q[i]); op 0 P + Q (complete), 1 the same mixed, 2 2P, 3 -P, 4 P - Q, 5 the same
 *   mixed, and the incomplete formulas inside their domain (P, Q finite, P != +-Q): 6 2P (Jacobian), 7 2P + Q (Jacobian
 *   doubling + mixed addition), 8 P + Q (XYZZ mixed), 9 P + Q (XYZZ affine + affine), 10 (k256; n a multiple of 64) 32 P of
 *   the wave's FIRST point in every lane, by five row-parallel complete doublings.
 * ECGPU_ERR_POINT: an input >= p / off the curve; ECGPU_ERR_SCALAR_RANGE: unknown op. */
int ecgpu_selftest_field(ecgpu_ctx *ctx, int curve, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
int ecgpu_selftest_point(ecgpu_ctx *ctx, int curve, int op, const uint8_t *p_xy, const uint8_t *p_inf, const uint8_t *q_xy,
                         const uint8_t *q_inf, size_t n, uint8_t *out_xy, uint8_t *out_inf);

/* Integer-VALU roof probe: runs a dependency-free v_mad_u64_u32 stream on every CU and returns
 * the measured 32x32->64 multiply-add rate in operations per second (SURVEY.md §8d "peak to
 * divide by").  `which` selects the instruction: 0 v_mad_u64_u32, 1 v_mul_lo_u32, 2 v_mul_hi_u32,
 * 3 v_add_u32, 4 v_lshl_add_u64, 5 v_mad_u32_u24, 6 v_fma_f64, 7 v_addc_co_u32 chain; 100 + i: exact
 * inline-asm instruction streams; 200: HBM gather probe — 2^20 lanes x 16 random 64-byte reads of the k256
 * comb table, returns BYTES per second (a known byte count in the fixed-base kernel's access pattern, used
 * to calibrate rocprofv3's FETCH_SIZE). */
int ecgpu_valu_probe(ecgpu_ctx *ctx, int which, double *ops_per_sec);

/* Milliseconds the device spent in the kernels of the last *_dev / host call on this context,
 * measured with HIP events on the context stream; `name` selects a stage:
 *   "total", "main" (the scalar-mul / bucket kernels), "normalize", "recode", "sort", "reduce". */
int ecgpu_last_timing(const ecgpu_ctx *ctx, const char *name, double *ms);

/* Per-call timing events on / off (default: on).  Every call brackets its kernels with HIP events so that ecgpu_last_timing
 * can report them; an event is a packet of its own on the stream, and on a queue of short batches they add up (measured: 8 us
 * of a 0.63 ms fixed-base batch).  on == 0: the calls record nothing and ecgpu_last_timing returns ECGPU_ERR_ARG for them. */
int ecgpu_set_timing(ecgpu_ctx *ctx, int on);

/* Library version string. */
const char *ecgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ECGPU_H */
