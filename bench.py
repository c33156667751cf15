#!/usr/bin/env python3
"""bench.py — throughput of the scalar-mul hot path on MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--only NAME] [--no-check] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM when the
timed region starts.  The top-level record is BASELINE.json's configs[1]; the other GPU configs are timed in the same run,
with the same K / W and the same fences.  Their headline numbers are ALSO top-level keys (`var_p256_value`,
`msm_k256_value`, `var_p384_value` + `_ms_per_step`, `_frac`, `_check`), their compact records are under "configs", and the
prose that used to be repeated per record is once under "notes": the line stays below 8 KB.

    fixed_k256  (top level, configs[1])  k256 fixed-base, 2^20 random scalars per GPU      -> scalar-muls/s
    var_p256    (configs[2])             p256 variable-base (ECDH shape), 2^20 pairs/GPU   -> scalar-muls/s
    msm_k256    (configs[3])             k256 MSM, 2^24 terms in total, sharded over GPUs  -> terms/s
    var_p384    (configs[4])             p384 variable-base, 2^20 pairs per GPU            -> scalar-muls/s
    msm_k256_2p21  (N = 1 only)          one GPU's share of configs[3] on 8 GPUs, timed right after configs[3] (same thermal state); its
                                         per-term rate over the 2^24 rate is the single-GPU projection of the 8-GPU efficiency
                                         (`projected_8gpu_efficiency`)
    ecdsa_p256, recover_k256             the signature callers of the path
    fixed_k256_ct, lincomb_ct_k256,      the names north_star uses in their CONSTANT-TIME meaning (`mul_by_generator`, `lincomb`, `P * k`): the
    var_p256_ct                          uniform-schedule kernels.  The headline and msm_k256 are the reference's *_vartime forms.
    msm_k256[_2p21]_lanes (N = 1)        the same MSMs with 3 / 2 in flight (ecgpu_set_msm_lanes); msm_k256_2p21_sharded_lanes: the share as a
                                         SHARDED step (parts / exchange / finish) with consecutive local halves on two rotating lanes;
                                         N > 1: msm_k256_sharded_lanes (the 2^24-term MSM itself in that form) and `per_rank` under msm_k256
                                         (every rank's local half, exchange alone, combining half: a scaling run that explains itself)
    fixed_k256_tier_ms (N = 1)           the headline batch on the narrower generator tables of the library's DEFAULT (adaptive) policy
    msm_k256.e2e_ms  (N = 1)             the 2^24-term MSM from HOST memory through ecgpu_msm (PCIe-inclusive; never `value`)
    group_msm_k256   (N > 1, rank 0)     the 2^24-term MSM through the single-process entry ecgpu_group_msm_dev

Every workload rotates over `--sets` (default 4) independently seeded input sets, one per step: the headline's 2^20 x 10 table
lines (671 MB per set) are not the same lines step after step, and the signature workloads hold 2^20 DISTINCT (z, r, s, Q)
tuples per set (2^16 nonces, each under 16 different keys and digests).  The check reads the set of the last step.
`table` (headline record) is the generator table in use: window bits, bytes of device memory, build time; the bench asks for
the widest table up front (ECGPU_TABLE_EAGER — a long-lived service; the library's default grows the table with use, include/ecgpu.h).
`fixed_k256_e2e_ms` is the same 2^20-scalar batch through the host-pointer entry ecgpu_batch_mul_base (32 MiB in, 65 MiB out
over PCIe from page-locked memory; never `value`).

`--only NAME` times a single workload as the top-level record (profiling runs; also var_k256, msm_p256, ecdsa_p256).
Batch workloads shard embarrassingly (weak scaling, no data-path collective).  The MSM shards its terms (strong
scaling) and has one exchange step: an RCCL all-gather of each rank's per-window partial sums, after which the window
sums over all ranks and the one Horner chain run on every rank (ecgpu_msm_parts_dev / ecgpu_msm_finish_dev,
elliptic-curves_amd/sharded.py).

`roofline` prices the dominant kernel against the integer-VALU issue roof (the path is neither HBM- nor
MFMA-bound).  The roof is one wave64 instruction slot per SIMD per cycle pair: on gfx950 a VOP3 / 64-bit instruction —
v_mad_u64_u32, the 32x32+64 multiply-add, among them — issues in 4 cycles and a 32-bit VOP1/VOP2 one in 2
(profiles/r01/isa_issue_rates.txt).  `peak` is that issue rate at the chip's peak engine clock: 256 CUs x 4 SIMDs x
16 lanes per cycle x 2.4 GHz (MI355X_MICROARCH.md) = 3.93e13 multiply-add slots per second; ecgpu_valu_probe measures
the same rate on this GPU in this run (`peak_probe`: every lane multiplying, the shader clock settles at ~2.1 GHz).
`achieved` is the EXECUTED work of the kernel in the same unit:
VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, committed under profiles/, constant for the seeded default
workload) x their mean issue cost in v_mad_u64_u32 slots (static ISA histogram of the kernel) x 64 lanes / the kernel's
average duration measured live with HIP events on the launch stream.  frac = achieved / peak is the utilisation of
the VALU issue roof; `mad_frac` is the share of it spent on multiply-adds proper.  The shader clock floats with the load
(the probe runs at ~2.1 GHz, the real kernels between 1.9 and 2.35), so two more views are printed: `frac_vs_probe`
(the same numerator over the probe's measured rate; it can exceed 1 when a kernel clocks higher than the probe) and
`frac_cycles_pmc` (executed issue cycles / GRBM_GUI_ACTIVE cycles of the committed PMC pass: no clock in it).  The reference algorithm's IMAD32
count of SURVEY.md §8d divided by the same time and peak is reported separately as `algorithmic_speedup` (it exceeds 1
when the GPU algorithm does less arithmetic per unit than the reference's).  The HBM view is under "hbm".
`cpu_baseline` times the oracle (a C restatement of the reference's own CPU algorithm, kind "port") on the host cores.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# reference field-multiplication count x canonical IMAD32 per field multiplication
# (F256 = 2*8^2 + 8 = 136, F384 = 2*12^2 + 12 = 300), and algorithmic HBM bytes per unit.
WORKLOADS = {
    # the headline is the VARIABLE-TIME generator multiplication (`mul_by_generator_vartime`, k256/src/arithmetic/mul.rs:205-232: a signed
    # comb whose table reads and additions follow the scalar); the constant-time `mul_by_generator` (:180-197) is fixed_k256_ct below
    "fixed_k256": dict(curve="k256", kind="fixed", n=1 << 20, metric="k256 fixed-base scalar-muls/sec (variable-time: mul_by_generator_vartime)",
                       unit="scalar-muls/s", imad_per_unit=812 * 136, bytes_per_unit=96, kernel="k_fixed_base<K256Params>", scaling="weak",
                       form="variable-time comb (ecgpu_batch_mul_base_dev = mul_by_generator_vartime)"),
    # `mul_by_generator` proper: the reference's constant-time schedule (k256 mul.rs:180-197 / primeorder basepoint.rs:82-99) —
    # every entry of every LUT read and one kept under a mask, complete formulas, no scalar-dependent branch or address
    "fixed_k256_ct": dict(curve="k256", kind="fixed", ct=True, n=1 << 20, metric="k256 fixed-base scalar-muls/sec (uniform schedule: mul_by_generator)",
                          unit="scalar-muls/s", imad_per_unit=812 * 136, bytes_per_unit=96, kernel="k_fixed_base_ct<K256Params>", scaling="weak"),
    # `LinearCombination::lincomb` in its constant-time meaning (k256 mul.rs:84-98, primeorder projective.rs:484-496): one
    # uniform-schedule multiplication per term + a tree of complete additions (ecgpu_lincomb_ct_dev); 2^20 terms
    "lincomb_ct_k256": dict(curve="k256", kind="lincomb_ct", n=1 << 20, metric="k256 lincomb terms/sec (uniform schedule: lincomb)",
                            unit="terms/s", imad_per_unit=960 * 136, bytes_per_unit=96, kernel="k_var_base_ct<K256Params>", scaling="weak"),
    "var_p256": dict(curve="p256", kind="var", n=1 << 20, metric="p256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=4336 * 136, bytes_per_unit=160, kernel="k_var_base<P256Params>", scaling="weak"),
    "var_k256": dict(curve="k256", kind="var", n=1 << 20, metric="k256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=1984 * 136, bytes_per_unit=160, kernel="k_var_base<K256Params>", scaling="weak"),
    # the uniform-schedule (constant-time) form of configs[2]: the reference's own `ProjectivePoint * Scalar` algorithm — table
    # [P..8P], 65 radix-16 digits, every table entry read and one kept under a mask, complete formulas (ecgpu_batch_mul_ct)
    "var_p256_ct": dict(curve="p256", kind="var", ct=True, n=1 << 20, metric="p256 variable-base scalar-muls/sec (uniform schedule)",
                        unit="scalar-muls/s", imad_per_unit=4336 * 136, bytes_per_unit=160, kernel="k_var_base_ct<P256Params>", scaling="weak"),
    "msm_p256": dict(curve="p256", kind="msm", n=1 << 24, metric="p256 MSM terms/sec", unit="terms/s",
                     imad_per_unit=int(16.06 * 11 * 136), bytes_per_unit=96 + 16 * 64, kernel="k_msm_accumulate<P256Params>",
                     scaling="strong"),
    # batch ECDSA verification: u1 G + u2 Q per signature = fixed-base + variable-base + 1 addition
    "ecdsa_p256": dict(curve="p256", kind="ecdsa", n=1 << 20, metric="p256 ECDSA verifications/sec", unit="verifications/s",
                       imad_per_unit=(962 + 4336 + 14) * 136, bytes_per_unit=160, kernel="k_var_base<P256Params>", scaling="weak"),
    # batch ECDSA public-key recovery (ecdsa `recover_from_prehash`, reference vectors k256/src/ecdsa.rs:190-262): per signature a
    # square root (decompression of R: ~268 M), r^-1, the 2-term `lincomb` (k256 mul.rs:112-163 with N = 2: 128 doublings + 160
    # additions = 2944 M) and the closing `verify_prehash` (GLV + wNAF a G + b P: ~1600 M)
    "recover_k256": dict(curve="k256", kind="recover", n=1 << 20, metric="k256 ECDSA public-key recoveries/sec", unit="recoveries/s",
                         imad_per_unit=(268 + 2944 + 1600) * 136, bytes_per_unit=97 + 64, kernel="k_var_base<K256Params>", scaling="weak"),
    "var_p384": dict(curve="p384", kind="var", n=1 << 20, metric="p384 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=6448 * 300, bytes_per_unit=240, kernel="k_var_base<P384Params>", scaling="weak"),
    "msm_k256": dict(curve="k256", kind="msm", n=1 << 24, metric="k256 MSM terms/sec", unit="terms/s",
                     # Pippenger, algorithmic: ceil(256/c) * (N + 2^c) mixed adds at 11 M, c = 16
                     imad_per_unit=int(16.06 * 11 * 136), bytes_per_unit=96 + 16 * 64, kernel="k_msm_accumulate<K256Params>",
                     scaling="strong"),
}
# one GPU's share of configs[3] on an 8-GPU node (2^24 / 8 terms): the part of the bucket method that does not shrink with n
# shows here; its per-term rate against the 2^24 rate is the single-GPU projection of the 8-GPU scaling efficiency
WORKLOADS["msm_k256_2p21"] = dict(WORKLOADS["msm_k256"], n=1 << 21, metric="k256 MSM terms/sec (2^21-term share)")
# the same two MSM workloads with SEVERAL MSMs in flight (ecgpu_set_msm_lanes on an asynchronous context: rotating internal streams
# and workspaces; N = 1 only): throughput of independent back-to-back MSMs, the time of a single one does not change
# (profiles/r03/msm_lanes.txt: two lanes are best at 2^21 terms, three at 2^24)
WORKLOADS["msm_k256_lanes"] = dict(WORKLOADS["msm_k256"], lanes=3, metric="k256 MSM terms/sec, three MSMs in flight")
WORKLOADS["msm_k256_2p21_lanes"] = dict(WORKLOADS["msm_k256_2p21"], lanes=2, metric="k256 MSM terms/sec (2^21-term share), two MSMs in flight")
# one GPU's share as a SHARDED step (ecgpu_msm_parts_dev / ecgpu_msm_finish_dev) with the local halves of consecutive MSMs on two rotating
# lanes: the exchange + combining half of step i under the accumulation of step i + 1 (N = 1: one rank, the "exchange" is the identity)
WORKLOADS["msm_k256_2p21_sharded_lanes"] = dict(WORKLOADS["msm_k256_2p21"], lanes=2, sharded=True,
                                                 metric="k256 MSM terms/sec (2^21-term share), sharded step on two lanes")
# N > 1: the whole 2^24-term MSM, sharded, consecutive MSMs on two lanes per GPU
WORKLOADS["msm_k256_sharded_lanes"] = dict(WORKLOADS["msm_k256"], lanes=2, sharded=True, metric="k256 MSM terms/sec, sharded steps on two lanes per GPU")
SEEDS = {"fixed_k256_ct": 12, "lincomb_ct_k256": 13, "msm_k256_2p21_sharded_lanes": 10, "msm_k256_sharded_lanes": 4, "msm_k256_lanes": 4, "msm_k256_2p21_lanes": 10, "var_p256_ct": 11, "msm_k256_2p21": 10, "fixed_k256": 2, "var_p256": 3, "msm_k256": 4, "var_p384": 5, "var_k256": 6, "ecdsa_p256": 7, "msm_p256": 8, "recover_k256": 9}
# BASELINE configs[2], [3], [4] beside the top-level configs[1], then the two signature workloads of SURVEY.md 8(f) (callers of the
# path: p256 verification, k256 public-key recovery) so that they are driver-timed too
DEFAULT_SUBS = ["var_p256", "msm_k256", "msm_k256_2p21", "var_p384", "ecdsa_p256", "recover_k256", "var_p256_ct", "fixed_k256_ct", "lincomb_ct_k256"]
NOMINAL_PEAK = 256 * 4 * 16 * 2.4e9   # IMAD32/s at the 2.4 GHz peak engine clock (the probe, all CUs multiplying, runs at ~2.1)
HBM_PEAK_GBPS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)
ROOFLINE_CONSTS = os.path.join(ROOT, "profiles", "roofline_consts.json")


def roofline_consts(kernel):
    """Executed-work constants of `kernel` for the seeded default workload, from the committed rocprofv3 PMC passes and
    the kernel's ISA histogram (tools/roofline_consts.py writes the file; PMC counters cannot be read from inside the
    timed process): {"units_per_launch", "insts_valu", "slots_per_inst", "mad_share", "fetch_bytes", "write_bytes", "source"}."""
    try:
        with open(ROOFLINE_CONSTS) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def device_random_scalars(torch, n, L, seed, device):
    """n uniformly random L-byte big-endian scalars < group order, generated on the GPU.  The orders of
    all three curves start with 0xffffffff, so clearing one bit of the (2^-32-rare) all-ones top word is
    enough to stay below n."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    b = torch.randint(0, 256, (n, L), dtype=torch.uint8, device=device, generator=g)
    top = (b[:, 0] == 255) & (b[:, 1] == 255) & (b[:, 2] == 255) & (b[:, 3] == 255)
    b[:, 3] = torch.where(top, torch.full_like(b[:, 3], 254), b[:, 3])
    return b.contiguous()


def device_dot_mod(torch, d_k, d_s, mod):
    """sum_i k_i s_i mod `mod` for two (n, L) big-endian byte tensors on the GPU: 16-bit limbs of k against 8-bit limbs
    of s in float64 matrix products — every partial sum is an integer below 2^53, so the result is exact."""
    n, L = d_k.shape
    total = 0
    step = 1 << 22
    for lo in range(0, n, step):
        kc = d_k[lo:lo + step].to(torch.float64)
        k16 = kc[:, 0::2] * 256.0 + kc[:, 1::2]
        m = (k16.T @ d_s[lo:lo + step].to(torch.float64)).cpu().numpy()
        for a in range(L // 2):
            wa = 16 * (L // 2 - 1 - a)
            for b in range(L):
                total += int(m[a, b]) << (wa + 8 * (L - 1 - b))
    return total % mod


def _sign_slice(job):
    """Worker (spawned process): s_i = k_j^-1 (z_i + r_j d_i) mod n for one slice of tuples, j = i mod m -> (s bytes, recovery ids)."""
    db, zb, rs, kinvs, odds, xhi, L, order, low_s, lo = job
    m = len(rs)
    half = order // 2
    out_s, out_id = bytearray(), bytearray()
    frm = int.from_bytes
    for i in range(len(db) // L):
        j = (lo + i) % m
        si = kinvs[j] * (frm(zb[i * L:(i + 1) * L], "big") + rs[j] * frm(db[i * L:(i + 1) * L], "big")) % order
        odd = odds[j]
        if low_s and si > half:                              # low-S form (k256 NORMALIZE_S): (r, -s) belongs to -R
            si, odd = order - si, odd ^ 1
        out_s += si.to_bytes(L, "big")
        out_id.append(odd | (2 if xhi[j] else 0))
    return bytes(out_s), bytes(out_id)


_SIGN_POOL = None


def sign_pool():
    """A pool of spawned (not forked: the parent holds a HIP context) worker processes for the host side of the signature inputs."""
    global _SIGN_POOL
    if _SIGN_POOL is None:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        world = int(os.environ.get("WORLD_SIZE", "1"))     # N ranks share the box's cores: each rank takes its share
        _SIGN_POOL = ProcessPoolExecutor(max_workers=max(1, min(16, host_cores() // max(1, world))), mp_context=mp.get_context("spawn"))
    return _SIGN_POOL


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the container's cgroup CPU quota (the GPU box shows
    256 logical CPUs but grants 16: /sys/fs/cgroup/cpu.max = "1600000 100000")."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(wl, cid, L, sample_scalars, sample_points, extra=None, target_wall=1.2):
    """Oracle ("port" of the reference's CPU algorithm) on the host cores: a single-thread pilot, then every core busy
    for about `target_wall` seconds (each thread works through a slice of the sample sized from the pilot rate)."""
    import oracle_lib
    oracle_lib.build()
    cores = host_cores()
    kind = wl["kind"]

    def run(lo, hi):
        s = sample_scalars[lo * L: hi * L]
        if kind == "fixed":
            oracle_lib.batch_mul_base(cid, s)
        elif kind == "var":
            oracle_lib.batch_mul(cid, s, sample_points[lo * 2 * L: hi * 2 * L])
        elif kind == "ecdsa":
            oracle_lib.ecdsa_verify(cid, s, extra[0][lo * L: hi * L], extra[1][lo * L: hi * L],
                                    sample_points[lo * 2 * L: hi * 2 * L])
        elif kind == "recover":
            oracle_lib.ecdsa_recover(cid, s, extra[0][lo * L: hi * L], extra[1][lo * L: hi * L], extra[2][lo:hi], True)
        else:       # the MSM's CPU counterpart: `lincomb_vartime`; for the uniform-schedule workload the constant-time `lincomb`
            oracle_lib.msm(cid, s, sample_points[lo * 2 * L: hi * 2 * L], vartime=kind != "lincomb_ct")

    avail = sample_scalars.size // L
    pilot = min(avail, 256 if kind != "fixed" else 2048)
    run(0, min(pilot, 32))                                   # first touch: tables, page faults
    t0 = time.perf_counter()
    run(0, pilot)
    single = pilot / (time.perf_counter() - t0)

    def all_threads(per_thread):
        per_thread = int(max(8, min(avail, per_thread)))
        starts = [(i * per_thread) % max(1, avail - per_thread + 1) for i in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda lo: run(lo, lo + per_thread), starts))
        return per_thread, time.perf_counter() - t0

    # a short all-thread probe measures how many cores the box really gives this process (the container may be limited to
    # fewer than os.cpu_count()); the timed pass is then sized for about `target_wall` seconds with every thread busy
    probe_units, probe_dt = all_threads(single * 0.04)
    rate = probe_units * cores / probe_dt
    per_thread, dt = all_threads(rate * target_wall / cores)
    total = per_thread * cores
    algo = {"fixed": "mul_by_generator (33/49-LUT basepoint table)", "var": "ProjectivePoint * Scalar (LUT + radix-16)",
            "msm": "lincomb_vartime (GLV + wNAF-5 Straus), one %d-term lincomb per thread" % per_thread,
            "lincomb_ct": "lincomb (constant-time GLV + radix-16 Straus), one %d-term lincomb per thread" % per_thread,
            "ecdsa": "verify_prehashed: s^-1, u1 G + u2 Q (mul_by_generator_and_mul_add_vartime), x mod n == r",
            "recover": "recover_from_prehash: decompress R, r^-1, lincomb(G, u1, R, u2), verify_prehash"}[kind]
    return {"value": total / dt, "unit": wl["unit"], "cores": cores, "kind": "port",
            "sample": "%d threads x %d units of the same seeded workload (slices of its first %d units), %s, oracle/ C restatement"
                      % (cores, per_thread, avail, algo),
            "wall_s": dt, "single_thread_value": single, "thread_scaling": total / dt / single}


def cpu_plumbing_record():
    """BASELINE.json configs[0]: k256 `ProjectivePoint::GENERATOR * random Scalar`, batch of 1024, on the CPU reference path —
    the oracle's restatement of mul_by_generator (k256/src/arithmetic/mul.rs:180-197), one thread, checked here against the
    pure-Python model (tests/pyec.py) on a few of the 1024 and against the group law (sum of the outputs == (sum of the
    scalars) G).  No GPU involved: `python bench.py --cpu-plumbing` runs anywhere; the default run files it under
    configs["cpu_k256_1024"]."""
    import oracle_lib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyec
    oracle_lib.build()
    c = pyec.CURVES["k256"]
    n, L = 1024, 32
    rng = np.random.default_rng(20260924)
    ks = [int.from_bytes(rng.bytes(40), "big") % (c.n - 1) + 1 for _ in range(n)]
    scal = np.frombuffer(b"".join(k.to_bytes(L, "big") for k in ks), np.uint8)
    oracle_lib.batch_mul_base(0, scal[:32 * L])                  # first touch (table build)
    t0 = time.perf_counter()
    out, inf = oracle_lib.batch_mul_base(0, scal)
    dt = time.perf_counter() - t0
    ok = not inf.any()
    for i in (0, 1, 511, 1023):
        P = pyec.mul(c, ks[i], pyec.G(c))
        ok = ok and bytes(out[i * 2 * L:(i + 1) * 2 * L]) == P[0].to_bytes(L, "big") + P[1].to_bytes(L, "big")
    tot, tf = oracle_lib.msm(0, np.tile(np.frombuffer((1).to_bytes(L, "big"), np.uint8), n), out, vartime=True)
    want, _ = oracle_lib.batch_mul_base(0, np.frombuffer((sum(ks) % c.n).to_bytes(L, "big"), np.uint8))
    ok = ok and tf == 0 and bytes(tot) == bytes(want)
    return {"metric": "k256 fixed-base scalar-muls/sec (CPU reference path)", "value": n / dt, "unit": "scalar-muls/s", "n_gpus": 0,
            "ms_per_step": dt * 1e3, "higher_is_better": True, "dtype": "u64 limbs (oracle, 5x52 field)", "data": "synthetic",
            "config": {"workload": "cpu_k256_1024", "curve": "k256", "units": n, "threads": 1,
                       "path": "oracle/ C restatement of mul_by_generator (kind \"port\")"},
            "check_vs_model": bool(ok)}


def group_msm_in_child(args, world, timeout=300.0):
    """Rank 0, N > 1: `bench.py --group-msm-child N` in a process of its own — all N GPUs from one process through
    ecgpu_group_msm_dev — with a deadline; -> its record, or {"skipped": reason}."""
    cmd = [sys.executable, os.path.abspath(__file__), "--group-msm-child", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.n:
        cmd += ["--n", str(args.n)]
    if args.no_check:
        cmd += ["--no-check"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")
           and not k.startswith("TORCHELASTIC")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"skipped": "ecgpu_group_msm child still running after %.0f s (killed)" % timeout}
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        tail = [ln for ln in (p.stderr or "").strip().splitlines() if ln.strip()]
        return {"skipped": "ecgpu_group_msm child exit %s: %s" % (p.returncode, (tail[-1] if tail else "no output")[:300])}
    return json.loads(lines[-1])


def group_msm_child_main(args):
    import torch
    b = Bench.__new__(Bench)
    b.torch, b.dist, b.args = torch, None, args
    b.world, b.rank, b.device = args.group_msm_child, 0, "cuda:0"
    torch.cuda.set_device(0)
    b.ecgpu = importlib.import_module("elliptic-curves_amd")
    b.eng = b.ecgpu.Engine(0)
    b.eng.set_stream(torch.cuda.current_stream().cuda_stream)
    b.eng.set_table_policy(b.ecgpu.TABLE_EAGER)
    try:
        return b.group_msm()
    finally:
        b.eng.close()


class Bench:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
            args.gpus = self.world
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a gfx950 GPU; there is no CPU path")
        # Dry-run knobs for a one-GPU box (tools/gpu_run.sh recipes `bench2` / `benchN`): ECGPU_BENCH_SHARE_GPU=1 puts every rank on
        # device 0 and ECGPU_BENCH_BACKEND=gloo moves the (tiny) exchange through the host, so that the N > 1 code path —
        # term shards, parts / finish around the all-gather, the collective check — can be exercised where RCCL cannot run
        # (it refuses two ranks on one device).  The driver's launch uses neither: one GPU per rank, RCCL.
        self.backend = os.environ.get("ECGPU_BENCH_BACKEND", "nccl")
        if os.environ.get("ECGPU_BENCH_SHARE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        self.device = "cuda:%d" % local_rank
        self.ecgpu = importlib.import_module("elliptic-curves_amd")
        self.exchange_kind, self.exchange_reason, self.group = "none", "one rank", None
        if self.world > 1:
            # The first contact with RCCL on N GPUs happens in the driver's ONE scaling run, so it must not be able to lose that
            # run: the control plane (barriers, max-over-ranks timing, the check's gathers) is gloo, RCCL is tried in a child
            # process per rank first and becomes the data path only if every canary and the group's first collective are
            # healthy; otherwise the 41 KiB records travel through gloo and the line says so (sharded.init_exchange).
            ex = self.ecgpu.init_exchange(torch, dist, local_rank, prefer=self.backend)
            self.exchange_kind, self.exchange_reason, self.group = ex.kind, ex.reason, ex.group
        self.eng = self.ecgpu.Engine(local_rank)
        self.eng.set_stream(torch.cuda.current_stream().cuda_stream)
        # steady-state throughput is what is measured: the widest generator table from the first call on (a long-lived service;
        # the library's default would grow the table with use, and the record says what the table costs: `table`)
        self.eng.set_table_policy(self.ecgpu.TABLE_EAGER)
        self.peak = None

    def fence(self):
        if self.world > 1:
            self.torch.cuda.synchronize()
            if self.group is not None:
                self.dist.barrier(group=self.group)          # RCCL: a device-side barrier (tens of microseconds)
            else:
                self.dist.barrier()                          # gloo (fallback / dry runs)
        self.torch.cuda.synchronize()

    def valu_peak(self):
        if self.peak is None:
            self.peak = self.eng.valu_probe(0)               # v_mad_u64_u32 lane-operations / s on this GPU
        return self.peak

    def make_inputs(self, name, kind, cid, L, n, seed):
        """One input set of a workload, on the device: {"scal", "pts", "s2", "r", "s", "recid"} (None where the kind has none)."""
        torch, eng, ecgpu, device = self.torch, self.eng, self.ecgpu, self.device
        inp = dict(scal=device_random_scalars(torch, n, L, seed, device), pts=None, s2=None, r=None, s=None, recid=None)
        if kind in ("var", "msm", "lincomb_ct"):
            inp["s2"] = device_random_scalars(torch, n, L, seed + 50, device)
            inp["pts"] = torch.empty((n, 2 * L), dtype=torch.uint8, device=device)
            torch.cuda.synchronize()
            eng.mul_by_generator_dev(cid, inp["s2"], n, inp["pts"], None)          # P_i = s_i * G (untimed setup)
        if kind in ("ecdsa", "recover"):
            # n DISTINCT valid signatures: tuple i is nonce k_(i mod m) (m = 2^16 nonce points R = k G computed here, their
            # inverses on the host) under a key d_i and a digest z_i of its own — Q_i = d_i G computed here,
            # s_i = k^-1 (z_i + r d_i) mod n on the host cores (worker processes).  What the verifier multiplies by — z/s, r/s,
            # Q — differs in every tuple; only r repeats.
            m = min(n, 1 << 16)
            d_d = device_random_scalars(torch, n, L, seed + 50, device)
            d_k = device_random_scalars(torch, m, L, seed + 51, device)
            d_Q = torch.empty((n, 2 * L), dtype=torch.uint8, device=device)
            d_R = torch.empty((m, 2 * L), dtype=torch.uint8, device=device)
            torch.cuda.synchronize()
            eng.mul_by_generator_dev(cid, d_d, n, d_Q, None)
            eng.mul_by_generator_dev(cid, d_k, m, d_R, None)
            torch.cuda.synchronize()
            order = ecgpu.GROUP_ORDERS[cid]
            kh, rx, ry = d_k.cpu().numpy(), d_R[:, :L].contiguous().cpu().numpy(), d_R[:, 2 * L - 1].cpu().numpy()
            rs, kinvs, odds, xhi = [], [], [], []
            for j in range(m):
                kj, xj = int.from_bytes(kh[j].tobytes(), "big") or 1, int.from_bytes(rx[j].tobytes(), "big")
                rs.append(xj % order); kinvs.append(pow(kj, -1, order)); odds.append(int(ry[j]) & 1); xhi.append(xj >= order)
            db, zb = d_d.cpu().numpy().tobytes(), inp["scal"].cpu().numpy().tobytes()
            pool = sign_pool()
            nw = pool._max_workers
            per = (n + nw - 1) // nw
            jobs = [(db[lo * L:(lo + per) * L], zb[lo * L:(lo + per) * L], rs, kinvs, odds, xhi, L, order, kind == "recover", lo)
                    for lo in range(0, n, per)]
            res = list(pool.map(_sign_slice, jobs))
            sb, ib = b"".join(r[0] for r in res), b"".join(r[1] for r in res)
            rb = b"".join(rs[i % m].to_bytes(L, "big") for i in range(m))
            reps = (n + m - 1) // m
            inp["r"] = torch.frombuffer(bytearray(rb), dtype=torch.uint8).reshape(m, L).to(device).repeat(reps, 1)[:n].contiguous()
            inp["s"] = torch.frombuffer(bytearray(sb), dtype=torch.uint8).reshape(n, L).to(device)
            inp["recid"] = torch.frombuffer(bytearray(ib), dtype=torch.uint8).to(device)
            inp["pts"] = d_Q
        return inp

    def run(self, name, cpu_leg):
        """Times workload `name`: W warm-up steps, then exactly K steps between fences; returns its record (rank 0) or None."""
        torch, dist, eng, ecgpu, args = self.torch, self.dist, self.eng, self.ecgpu, self.args
        world, rank, device = self.world, self.rank, self.device
        wl = WORKLOADS[name]
        cid = ecgpu.CURVE_IDS[wl["curve"]]
        L = ecgpu.FIELD_BYTES[cid]
        kind = wl["kind"]
        n_total = args.n or wl["n"]
        if kind == "msm":
            lo, hi = ecgpu.shard_range(n_total, rank, world)
            n = hi - lo
            eng.set_msm_window(args.window if args.window else 0)
        else:
            n = n_total
            if args.window and kind == "fixed":
                eng.set_base_window(cid, args.window)

        # ---- synthetic inputs, resident in HBM before the timed region: `nsets` independently seeded sets, one per step in turn ----
        nsets = max(1, args.sets)
        if kind in ("ecdsa", "recover") and world > 1:
            nsets = min(nsets, 2)                      # (the host side of 2^20 signatures per set and rank: keep an N-rank run's setup short)
        sets = [self.make_inputs(name, kind, cid, L, n, 0xEC000000 + SEEDS[name] + 1000 * rank + 7919 * j) for j in range(nsets)]
        d_scal = d_pts = d_s2 = d_r = d_s = d_recid = None      # (bound to the set of the last step before the check)
        d_ok = torch.zeros((n + 16,), dtype=torch.uint8, device=device) if kind in ("ecdsa", "recover") else None
        n_out = 1 if kind in ("msm", "lincomb_ct") else n
        d_out = torch.empty((n_out, 2 * L), dtype=torch.uint8, device=device)
        d_inf = torch.empty((max(n_out, 16),), dtype=torch.uint8, device=device)
        # The MSM as a SHARDED step — local half (ecgpu_msm_parts_dev), ONE exchange (RCCL all-gather of the per-window partial sums
        # over xGMI), combining half on every rank (ecgpu_msm_finish_dev): always with N > 1 ranks, and at N = 1 for the workloads
        # that time one rank's share in that form (`sharded`: the exchange of one rank is the identity).
        sharded = kind == "msm" and (world > 1 or bool(wl.get("sharded")))
        lanes = int(wl.get("lanes", 1)) if kind == "msm" else 1
        lane_out = [(d_out, d_inf)] + [(torch.empty_like(d_out), torch.empty_like(d_inf)) for _ in range(lanes - 1)]
        exchanges, plan_terms = [], 0
        if sharded:
            plan_terms = (n_total + world - 1) // world              # the largest shard: every rank plans the same windows
            nbytes = eng.msm_parts_bytes(cid, plan_terms)
            for _ in range(lanes):                                   # a parts record (and its gathered form) per lane
                if world > 1:
                    exchanges.append(ecgpu.RecordExchange(torch, dist, nbytes, device, group=self.group))
                else:
                    exchanges.append(ecgpu.LocalRecord(torch, nbytes, device))
        torch.cuda.synchronize()     # inputs were written on torch's stream; make sure they are there whatever stream the engine uses

        main_ms, stages = [], {}
        # fixed / variable base, and the sharded MSM of an N-rank job: the calls are queued back to back (ecgpu_set_async) and the queue
        # is drained inside the timed region — no rank's host stands between two steps —; the per-call HIP events are then read for
        # the last timed step.  The other kinds keep the synchronous calls and read every step's events.
        queued = (kind in ("fixed", "var") or (sharded and world > 1)) and not args.sync_calls
        if lanes > 1:
            queued = True
        nstep = [0]
        pend = []                                      # sharded steps on lanes whose combining half is still to be queued: (lane, input set)

        def read_events():
            main_ms.append(eng.last_timing("accumulate" if kind == "msm" else "main") or 0.0)
            for st in ("sort", "accumulate", "reduce", "normalize", "main", "total", "prepare", "finish", "tree", "combine"):
                v = eng.last_timing(st)
                if v is not None:
                    stages.setdefault(st, []).append(v)

        written = {}                                   # index of an output buffer -> the input set of the last step that wrote it

        def combine_pending():
            """exchange + combining half of the oldest local half in flight, on the engine's stream (= torch's current stream: the
            collective is ordered behind ecgpu_msm_parts_join_dev and before ecgpu_msm_finish_dev on the device, no host wait)"""
            b, j = pend.pop(0)
            ex = exchanges[b]
            eng.msm_parts_join_dev(ex.mine)
            eng.msm_finish_dev(cid, ex.gather(consumer_on_current_stream=True), world, plan_terms, *lane_out[b])
            written[b] = j

        def step(read=False):
            j = nstep[0] % nsets
            nstep[0] += 1
            inp = sets[j]
            k_, p_ = inp["scal"], inp["pts"]
            buf = nstep[0] % lanes if lanes > 1 else 0
            if not (sharded and lanes > 1):
                written[buf] = j
            if kind == "fixed":
                eng.mul_by_generator_dev(cid, k_, n, d_out, d_inf, constant_time=bool(wl.get("ct")))
            elif kind == "var":
                eng.mul_dev(cid, k_, p_, None, n, d_out, d_inf, constant_time=bool(wl.get("ct")))
            elif kind == "lincomb_ct":
                eng.lincomb_ct_dev(cid, k_, p_, None, n, d_out, d_inf)
            elif kind == "ecdsa":
                eng.ecdsa_verify_dev(cid, k_, inp["r"], inp["s"], p_, n, False, d_ok)
            elif kind == "recover":
                eng.ecdsa_recover_dev(cid, k_, inp["r"], inp["s"], inp["recid"], n, True, d_out, d_ok)
            elif not sharded:
                eng.lincomb_dev(cid, k_, p_, None, n, *lane_out[buf])     # (several MSMs in flight: each writes buffers of its own)
            elif lanes > 1:
                # local half of step i on lane i % lanes; THEN the exchange + combining half of step i - 1 on the engine's stream,
                # beside it (include/ecgpu.h, ecgpu_msm_parts_join_dev)
                eng.msm_parts_dev(cid, k_, p_, None, n, plan_terms, exchanges[buf].mine)
                pend.append((buf, j))
                if len(pend) > 1:
                    combine_pending()
                return
            else:
                ex = exchanges[0]
                eng.msm_parts_dev(cid, k_, p_, None, n, plan_terms, ex.mine)
                if read:
                    read_events()                      # (the local half's events: the combining half below replaces them)
                eng.msm_finish_dev(cid, ex.gather(consumer_on_current_stream=True), world, plan_terms, d_out, d_inf)
                return
            if read:
                read_events()

        def flush():
            while pend:
                combine_pending()

        for _ in range(args.warmup):
            step()
        flush()
        if queued:
            eng.set_async(True)
        if lanes > 1:
            eng.set_msm_lanes(lanes)
            for _ in range(lanes):                       # the lanes' streams and workspaces exist before the clock starts
                step()
            flush()
            eng.synchronize()
        self.fence()
        # queued batches: the per-call timing events (three packets of their own per batch on the stream) are recorded for the
        # LAST timed step only — that step's kernel durations are what roofline.kernel_ms reads; the steps before it put
        # nothing but their kernels on the stream (ecgpu_set_timing; measured: 8-10 us of a 0.63 ms fixed-base batch)
        events_last_only = queued and lanes == 1
        if events_last_only:
            eng.set_timing(False)
        t0 = time.perf_counter()
        for i in range(args.steps):
            if events_last_only and i == args.steps - 1:
                eng.set_timing(True)
            step(read=not queued)
        flush()
        if queued:
            eng.synchronize()            # waits for the queue and raises if any queued batch failed its input checks
        self.fence()
        elapsed = time.perf_counter() - t0
        if queued:
            if not sharded:
                read_events()
            elif lanes > 1:                              # the last lane's accumulation kernel (ecgpu_last_timing reads the lane's events)
                main_ms.append(eng.last_timing("accumulate") or 0.0)
            if lanes > 1:
                eng.set_msm_lanes(1)
            eng.set_async(False)
            if lanes == 1:
                # the queued steps leave ONE sample of the kernel's duration (the HIP events of the last step); five more steps,
                # outside the timed region and synchronous, give kernel_ms a minimum and a mean to stand on
                eng.set_timing(True)
                for _ in range(5):
                    step(read=True)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)           # (the control plane is gloo: a host tensor)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per_rank = self.per_rank_stages(cid, sets[0], n, plan_terms, exchanges[0], d_out, d_inf) if sharded and world > 1 and lanes == 1 else None

        units_per_step = n_total if kind == "msm" else n * world
        value = units_per_step * args.steps / elapsed
        last = sets[written[0]]                        # the inputs of the step whose output d_out / d_ok hold now
        d_scal, d_pts, d_s2, d_r, d_s, d_recid = (last[k] for k in ("scal", "pts", "s2", "r", "s", "recid"))

        # ---- parity check of the last timed step (every rank takes part in the MSM's collective sum) ----
        ok = None
        if not args.no_check:
            if kind in ("msm", "lincomb_ct"):
                part = device_dot_mod(torch, d_scal, d_s2, ecgpu.GROUP_ORDERS[cid])
                if world > 1 and kind == "msm":
                    parts = [None] * world
                    dist.all_gather_object(parts, part)
                    part = sum(parts) % ecgpu.GROUP_ORDERS[cid]
            if rank == 0:
                import oracle_lib
                oracle_lib.build()
                if kind in ("msm", "lincomb_ct"):
                    # P_i = s_i G  =>  sum_i k_i P_i == (sum_i k_i s_i mod n) G, whatever the term count
                    w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer(part.to_bytes(L, "big"), np.uint8))
                    ok = bytes(w) == bytes(d_out[0].cpu().numpy()) and int(wf[0]) == int(d_inf[0].item())
                    if n_total <= (1 << 12) and world == 1:
                        w2, wf2 = oracle_lib.msm(cid, d_scal.cpu().numpy().reshape(-1), d_pts.cpu().numpy().reshape(-1), vartime=True)
                        ok = ok and bytes(w2) == bytes(w) and wf2 == int(wf[0])
                    if kind == "lincomb_ct":
                        # and the first 256 terms against the oracle's constant-time `lincomb` driver (k256 mul.rs:84-98), byte for byte
                        m = min(n, 256)
                        d_o2 = torch.empty((1, 2 * L), dtype=torch.uint8, device=device)
                        d_i2 = torch.zeros((16,), dtype=torch.uint8, device=device)
                        torch.cuda.synchronize()
                        eng.lincomb_ct_dev(cid, d_scal[:m].contiguous(), d_pts[:m].contiguous(), None, m, d_o2, d_i2)
                        w2, wf2 = oracle_lib.msm(cid, d_scal[:m].cpu().numpy().reshape(-1), d_pts[:m].cpu().numpy().reshape(-1), vartime=False)
                        ok = ok and bytes(w2) == bytes(d_o2[0].cpu().numpy()) and int(wf2) == int(d_i2[0].item())
                elif kind == "ecdsa":
                    m = min(n, 256)
                    w = oracle_lib.ecdsa_verify(cid, d_scal[:m].cpu().numpy().reshape(-1), d_r[:m].cpu().numpy().reshape(-1),
                                                d_s[:m].cpu().numpy().reshape(-1), d_pts[:m].cpu().numpy().reshape(-1))
                    ok = bool(w.all()) and bool(d_ok[:n].all().item())       # every synthetic signature is valid
                elif kind == "recover":
                    m = min(n, 256)
                    w, wok = oracle_lib.ecdsa_recover(cid, d_scal[:m].cpu().numpy().reshape(-1), d_r[:m].cpu().numpy().reshape(-1),
                                                      d_s[:m].cpu().numpy().reshape(-1), d_recid[:m].cpu().numpy(), True)
                    # every synthetic signature recovers to its signer's key: ALL n keys against the keys the signatures were made for
                    ok = bool(wok.all()) and bytes(w) == bytes(d_out[:m].cpu().numpy().reshape(-1)) and bool(d_ok[:n].all().item()) \
                        and bool(torch.equal(d_out, d_pts))
                else:
                    # (1) 256 outputs spread over the batch, byte for byte against the oracle
                    idx = torch.arange(0, n, max(1, n // 256), device=device)[:256]
                    got = d_out[idx].cpu().numpy().reshape(-1)
                    sh = d_scal[idx].cpu().numpy().reshape(-1)
                    if kind == "fixed":
                        w, _ = oracle_lib.batch_mul_base(cid, sh)
                    else:
                        w, _ = oracle_lib.batch_mul(cid, sh, d_pts[idx].cpu().numpy().reshape(-1))
                    ok = bytes(w) == bytes(got)
                    # (2) EVERY output, through the group law: sum_i out_i == (sum_i k_i [s_i]) G — one wrong element
                    # anywhere in the batch changes the sum (a checksum of the whole launch, exact)
                    order = ecgpu.GROUP_ORDERS[cid]
                    ones = torch.zeros((n, L), dtype=torch.uint8, device=device)
                    ones[:, L - 1] = 1
                    tot = device_dot_mod(torch, d_scal, d_s2 if kind == "var" else ones, order)
                    d_sum = torch.empty((1, 2 * L), dtype=torch.uint8, device=device)
                    d_sf = torch.zeros((16,), dtype=torch.uint8, device=device)
                    torch.cuda.synchronize()
                    eng.point_sum_dev(cid, d_out, d_inf, n, d_sum, d_sf)
                    w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer(tot.to_bytes(L, "big"), np.uint8))
                    ok = ok and bytes(w) == bytes(d_sum[0].cpu().numpy()) and int(wf[0]) == int(d_sf[0].item())
        if rank != 0:
            return None

        kernel_ms = float(np.mean(main_ms)) if main_ms else None
        kernel_ms_min = float(np.min(main_ms)) if main_ms else None
        peak = self.valu_peak()
        ksec = kernel_ms * 1e-3 if kernel_ms else None
        rc = roofline_consts(wl["kernel"])
        achieved = mad = traffic = frac_cycles = clock_kernel = None
        basis = "no committed PMC constants for this kernel"
        if rc and ksec:
            scale = n / rc["units_per_launch"]
            insts = rc["insts_valu"] * scale
            achieved = insts * rc["slots_per_inst"] * 64 / ksec
            mad = insts * rc["mad_share"] * 64 / ksec
            if rc.get("gui_cycles"):
                # clock-independent view: issue cycles executed / cycles the 1024 SIMDs had under the profiler
                # (GRBM_GUI_ACTIVE sums the 8 XCDs), and the shader clock that cycle count implies for the live duration
                # (both sides from the SAME committed PMC pass)
                frac_cycles = rc["insts_valu"] * rc["slots_per_inst"] * 4 / (rc["gui_cycles"] / 8 * 1024)
                clock_kernel = rc["gui_cycles"] / 8 * scale / ksec / 1e9
            traffic = (rc["fetch_bytes"] + rc["write_bytes"]) * scale if rc.get("fetch_bytes") is not None else None
            basis = rc["source"] + ("" if scale == 1 and not args.window else " (scaled from %d units per launch)" % rc["units_per_launch"])
        hbm_gbps = wl["bytes_per_unit"] * n / ksec / 1e9 if ksec else None
        rec = {
            "metric": wl["metric"], "value": value, "unit": wl["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": wl["scaling"], "vs_baseline": None, "dtype": "u32 limbs (v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": name, "curve": wl["curve"], "units_per_gpu": n, "units_total": units_per_step,
                       "window_bits": args.window or "default", "parallelism": "shard%d" % world,
                       **({"exchange": self.exchange_kind, "exchange_reason": self.exchange_reason} if world > 1 else {}),
                       **({"dry_run": "ranks share one GPU, %s exchange" % self.backend} if os.environ.get("ECGPU_BENCH_SHARE_GPU") else {})},
            "calls": "%d %s in flight (ecgpu_set_async + ecgpu_set_msm_lanes), drained inside the timed region; kernel_ms = the last "
                     "accumulation kernel with the other lanes' kernels beside it" % (lanes, "sharded steps" if sharded else "MSMs") if lanes > 1
                     else "queued (ecgpu_set_async), drained inside the timed region; kernel_ms = mean (kernel_ms_min: minimum) over the HIP "
                          "events of the last timed step and of five synchronous steps after the timed region" if queued
                     else "synchronous, kernel_ms averaged over the HIP events of every timed step",
            "roofline": {"bound": "valu-int", "kernel": wl["kernel"], "kernel_ms": kernel_ms, "kernel_ms_min": kernel_ms_min,
                         "kernel_ms_samples": len(main_ms),
                         # the WHOLE step priced like SURVEY 8d prices the path (reference algorithm's multiply-adds per unit x units
                         # / step time / peak): what the workload, not its dominant kernel, makes of the issue roof
                         "workload_frac_8d": wl["imad_per_unit"] * n / (elapsed / args.steps) / NOMINAL_PEAK,
                         "achieved": achieved / 1e12 if achieved else None, "peak": NOMINAL_PEAK / 1e12, "unit": "TIMAD32-slots/s",
                         "frac": achieved / NOMINAL_PEAK if achieved else None,
                         "mad_frac": mad / NOMINAL_PEAK if mad else None,
                         "peak_probe": peak / 1e12, "frac_vs_probe": achieved / peak if achieved else None,
                         "frac_cycles_pmc": frac_cycles, "clock_ghz_kernel": clock_kernel, "clock_ghz_probe": peak / (1024 * 16) / 1e9,
                         "traffic": traffic, "traffic_unit": "bytes/launch", "basis": basis,
                         "algorithmic_imad32_per_unit": wl["imad_per_unit"], "units_per_launch": n,
                         "algorithmic_speedup": wl["imad_per_unit"] * n / ksec / NOMINAL_PEAK if ksec else None,
                         "peak_source": "256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md; v_mad_u64_u32 issues at full rate: "
                                        "peak_probe = ecgpu_valu_probe measured in this run at the clock the probe reaches)",
                         "hbm": {"achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": hbm_gbps / HBM_PEAK_GBPS if hbm_gbps else None,
                                 "algorithmic_bytes_per_unit": wl["bytes_per_unit"],
                                 "measured_frac": traffic / ksec / 1e9 / HBM_PEAK_GBPS if traffic and ksec else None}},
            "stage_ms": {k: float(np.mean(v)) for k, v in stages.items()},
            "check_vs_oracle": ok,
        }
        rec["config"]["input_sets"] = nsets
        if wl.get("form"):
            rec["config"]["form"] = wl["form"]
        if per_rank:
            rec["per_rank"] = per_rank
        if kind in ("fixed", "ecdsa", "recover"):
            ti = eng.base_table_info(cid)               # the generator table these steps read (built before the timed region)
            rec["table"] = {"window_bits": ti["window_bits"], "bytes": ti["bytes"], "build_ms": round(ti["build_ms"], 2), "policy": "eager"}
        if cpu_leg:
            ns = min(n, 1 << 17 if kind in ("fixed", "msm", "lincomb_ct") else 1 << 14)
            s_host = d_scal[:ns].cpu().numpy().reshape(-1)
            p_host = d_pts[:ns].cpu().numpy().reshape(-1) if d_pts is not None else None
            extra = (d_r[:ns].cpu().numpy().reshape(-1), d_s[:ns].cpu().numpy().reshape(-1)) if kind in ("ecdsa", "recover") else None
            if kind == "recover":
                extra = extra + (d_recid[:ns].cpu().numpy(),)
            rec["cpu_baseline"] = cpu_baseline(wl, cid, L, s_host, p_host, extra)
        return rec

    def per_rank_stages(self, cid, inp, n, plan_terms, ex, d_out, d_inf):
        """N > 1, after the timed region: what every rank's share of the sharded MSM is made of, so that a scaling run explains itself
        — three synchronous, separately timed steps per rank: the local half (host clock around the call + its stage events), the
        exchange alone (ranks aligned by a barrier first, so that the figure is the collective and not the wait for the slowest rank),
        the combining half.  Minimum of the three; every rank's record travels to rank 0 over the gloo control plane."""
        torch, dist, eng, world = self.torch, self.dist, self.eng, self.world
        best = {}
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.msm_parts_dev(cid, inp["scal"], inp["pts"], None, n, plan_terms, ex.mine)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st = {k: eng.last_timing(k) for k in ("sort", "accumulate", "reduce")}
            dist.barrier()                                   # (gloo: the control plane)
            t2 = time.perf_counter()
            allrec = ex.gather(consumer_on_current_stream=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            eng.msm_finish_dev(cid, allrec, world, plan_terms, d_out, d_inf)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            cur = {"parts_ms": (t1 - t0) * 1e3, "exchange_us": (t3 - t2) * 1e6, "finish_ms": (t4 - t3) * 1e3,
                   **{k + "_ms": v for k, v in st.items() if v is not None}}
            for k, v in cur.items():
                best[k] = min(best.get(k, v), v)
        mine = {"rank": self.rank, "terms": n, **{k: float("%.4g" % v) for k, v in best.items()}}
        every = [None] * world
        dist.all_gather_object(every, mine)
        return every

    def e2e_msm(self, name="msm_k256"):
        """SURVEY.md 8d, config 4 "end-to-end incl. PCIe": the same 2^24-term problem handed over in (page-locked) HOST memory
        to the host-pointer entry point ecgpu_msm — upload of 1.5 GB in chunks under the compute of the previous chunk, partial
        MSMs, point sum, download of 65 bytes.  One warm-up call, then the best of two; never `value`."""
        torch, eng, ecgpu = self.torch, self.eng, self.ecgpu
        wl = WORKLOADS[name]
        cid = ecgpu.CURVE_IDS[wl["curve"]]
        L = ecgpu.FIELD_BYTES[cid]
        n = self.args.n or wl["n"]
        seed = 0xEC000000 + SEEDS[name]
        d_scal = device_random_scalars(torch, n, L, seed, self.device)
        d_s2 = device_random_scalars(torch, n, L, seed + 50, self.device)
        d_pts = torch.empty((n, 2 * L), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize()
        eng.mul_by_generator_dev(cid, d_s2, n, d_pts, None)
        torch.cuda.synchronize()
        h_s, h_p = eng.host_alloc(n * L), eng.host_alloc(n * 2 * L)
        try:
            torch.from_numpy(h_s).copy_(d_scal.view(-1))
            torch.from_numpy(h_p).copy_(d_pts.view(-1))
            torch.cuda.synchronize()
            best, out = None, None
            for it in range(3):
                t0 = time.perf_counter()
                out = eng.lincomb(cid, h_s, h_p)
                dt = time.perf_counter() - t0
                if it and (best is None or dt < best):
                    best = dt
            ok = None
            if not self.args.no_check:
                import oracle_lib
                oracle_lib.build()
                tot = device_dot_mod(torch, d_scal, d_s2, ecgpu.GROUP_ORDERS[cid])
                w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer(tot.to_bytes(L, "big"), np.uint8))
                ok = bytes(w) == bytes(out[0]) and int(wf[0]) == int(out[1])
        finally:
            eng.host_free(h_s)
            eng.host_free(h_p)
        return {"e2e_ms": best * 1e3, "e2e_value": n / best, "e2e_check": ok,
                "e2e_bytes_h2d": n * 3 * L}

    def e2e_fixed(self, name="fixed_k256"):
        """SURVEY.md 8d, config 2 with the boundary's host buffers: the same 2^20-scalar batch through the host-pointer entry
        ecgpu_batch_mul_base — 32 MiB of scalars up, 64 MiB of points + 1 MiB of flags down, page-locked memory, the library's
        chunked upload / compute / download pipeline.  One warm-up call, then the best of two; never `value`."""
        torch, eng, ecgpu = self.torch, self.eng, self.ecgpu
        wl = WORKLOADS[name]
        cid = ecgpu.CURVE_IDS[wl["curve"]]
        L = ecgpu.FIELD_BYTES[cid]
        n = self.args.n or wl["n"]
        d_scal = device_random_scalars(torch, n, L, 0xEC000000 + SEEDS[name] + 31, self.device)
        h_s, h_o, h_i = eng.host_alloc(n * L), eng.host_alloc(n * 2 * L), eng.host_alloc(n)
        try:
            torch.from_numpy(h_s).copy_(d_scal.view(-1))
            torch.cuda.synchronize()
            best = None
            for it in range(3):
                t0 = time.perf_counter()
                eng.mul_by_generator(cid, h_s, out=h_o, inf=h_i)
                dt = time.perf_counter() - t0
                if it and (best is None or dt < best):
                    best = dt
            ok = None
            if not self.args.no_check:
                import oracle_lib
                oracle_lib.build()
                idx = np.arange(0, n, max(1, n // 256))[:256]
                w, wf = oracle_lib.batch_mul_base(cid, h_s.reshape(n, L)[idx].reshape(-1).copy())
                ok = bytes(w) == bytes(h_o.reshape(n, 2 * L)[idx].reshape(-1)) and bytes(wf) == bytes(h_i[idx])
        finally:
            for h in (h_s, h_o, h_i):
                eng.host_free(h)
        return {"e2e_ms": best * 1e3, "e2e_value": n / best, "e2e_check": ok, "e2e_bytes": n * (3 * L + 1)}

    def group_msm(self, name="msm_k256"):
        """N > 1 only, rank 0 only (the other ranks wait at the barrier that follows): the SAME 2^24-term problem through the
        single-process entry ecgpu_group_msm_dev — one context + worker thread per GPU inside this process, the shards
        resident on their devices, the library's own exchange (RCCL ncclAllGather via dlopen, or peer copies) — what a Rust
        caller of `lincomb` reaches the node with.  Timed like a step: K synchronous calls between fences."""
        torch, ecgpu, args = self.torch, self.ecgpu, self.args
        wl = WORKLOADS[name]
        cid = ecgpu.CURVE_IDS[wl["curve"]]
        L = ecgpu.FIELD_BYTES[cid]
        n_total = args.n or wl["n"]
        world = self.world
        if torch.cuda.device_count() < world:
            return {"skipped": "this process sees %d devices" % torch.cuda.device_count()}
        try:
            grp = ecgpu.Group(list(range(world)))
        except ecgpu.EcgpuError as e:
            return {"skipped": "ecgpu_group_init: %s" % e}
        try:
            ds, dp, ns, dot = [], [], [], 0
            for r in range(world):
                lo, hi = ecgpu.shard_range(n_total, r, world)
                dev = "cuda:%d" % r
                seed = 0xEC000000 + SEEDS[name] + 1000 * r
                k = device_random_scalars(torch, hi - lo, L, seed, dev)
                s2 = device_random_scalars(torch, hi - lo, L, seed + 50, dev)
                pts = torch.empty((hi - lo, 2 * L), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)
                e = self.eng if r == 0 else ecgpu.Engine(r)
                e.mul_by_generator_dev(cid, s2, hi - lo, pts, None)
                torch.cuda.synchronize(dev)
                if r:
                    e.close()
                if not args.no_check:
                    dot += device_dot_mod(torch, k, s2, ecgpu.GROUP_ORDERS[cid])
                ds.append(k); dp.append(pts); ns.append(hi - lo)
            for _ in range(max(1, args.warmup)):
                out, inf = grp.lincomb_dev(cid, ds, dp, ns)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out, inf = grp.lincomb_dev(cid, ds, dp, ns)
            dt = (time.perf_counter() - t0) / args.steps
            ok = None
            if not args.no_check:
                import oracle_lib
                oracle_lib.build()
                w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer((dot % ecgpu.GROUP_ORDERS[cid]).to_bytes(L, "big"), np.uint8))
                ok = bytes(w) == bytes(out) and int(wf[0]) == int(inf)
            return {"value": n_total / dt, "unit": "terms/s", "ms_per_step": dt * 1e3, "exchange": grp.exchange,
                    "exchange_reason": grp.exchange_reason, "check_vs_oracle": ok,
                    "calls": "synchronous ecgpu_group_msm_dev from one process, result (65 bytes) downloaded inside the call"}
        finally:
            grp.close()

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()
        self.eng.close()


NOTES = {
    "roofline": "bound valu-int (SURVEY 8d). peak = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz v_mad_u64_u32 slots/s; achieved = SQ_INSTS_VALU per launch "
                "(rocprofv3 PMC, profiles/roofline_consts.json) x issue slots per instruction (ISA histogram) x 64 / kernel_ms (HIP events, this run); "
                "frac = achieved/peak; fvp = achieved / this GPU's probed issue rate (a slow box lowers frac, not fvp); wfrac = 8d multiply-adds x "
                "units / STEP time / peak; cyc = issue cycles / GRBM_GUI_ACTIVE (PMC); clk = PMC cycles / kernel_ms; algo_x = 8d IMAD32 / kernel time "
                "/ peak; traffic = FETCH_SIZE + WRITE_SIZE bytes per launch; units: the metric's, per second",
    "cpu": "oracle/ C restatement of the reference's CPU algorithm (kind port; no rustc here), all granted cores ~1.2 s; one = 1 thread",
    "check": "last timed step vs the oracle: batches 256 sampled outputs byte for byte AND sum of ALL outputs == (sum k_i [s_i]) G; MSMs == "
             "(sum k_i s_i mod n) G (lincomb_ct also 256 terms vs the oracle's constant-time lincomb); signatures: every verdict / key",
    "calls": "batches (and sharded MSM steps at N > 1) queued (ecgpu_set_async), drained inside the timed region; others synchronous; *_lanes: 2 "
             "(2^21) / 3 (2^24) MSMs in flight; *_sharded_lanes: parts / exchange / finish of consecutive MSMs on 2 rotating lanes",
}


def compact(r):
    """A sub-record for the one-line output: numbers only (the prose lives once in NOTES), ~400 bytes each — the driver keeps
    the last 8 KB of the line."""
    rf = r.get("roofline", {})
    g = lambda v, d=4: None if v is None else float(("%%.%dg" % d) % v)
    out = {"metric": r["metric"], "value": g(r["value"], 5), "ms_per_step": g(r["ms_per_step"], 5),
           "kernel": rf.get("kernel"), "kernel_ms": g(rf.get("kernel_ms")), "kmin": g(rf.get("kernel_ms_min")), "frac": g(rf.get("frac")),
           "fvp": g(rf.get("frac_vs_probe")), "wfrac": g(rf.get("workload_frac_8d")), "cyc": g(rf.get("frac_cycles_pmc")),
           "clk": g(rf.get("clock_ghz_kernel"), 3), "mad_frac": g(rf.get("mad_frac")), "algo_x": g(rf.get("algorithmic_speedup")),
           "traffic": g(rf.get("traffic")),
           "stage_ms": {k: g(v, 3) for k, v in r.get("stage_ms", {}).items() if k not in ("main", "total")},
           "check": r.get("check_vs_oracle")}
    cb = r.get("cpu_baseline")
    if cb:
        out["cpu"] = {"value": g(cb["value"]), "cores": cb["cores"], "one": g(cb["single_thread_value"]), "kind": cb["kind"]}
    for k in ("e2e_ms", "e2e_value", "e2e_check", "share_of_2p24_rate", "projected_8gpu_efficiency", "per_rank"):
        if k in r:
            out[k] = g(r[k]) if isinstance(r[k], float) else r[k]
    return out


def cargo_probe():
    """"re-probe cargo on the benchmark box; if present, time the real crates".  The workspace needs ~150
    crates.io packages and there is no network, so a present cargo is reported, not used."""
    import shutil
    import subprocess
    exe = shutil.which("cargo")
    if not exe:
        return None
    try:
        return subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=20).stdout.strip() or exe
    except Exception:
        return exe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", "--workload", dest="only", default=None, choices=sorted(WORKLOADS),
                    help="time this workload alone as the top-level record (default: fixed_k256 + the other GPU configs as sub-records)")
    ap.add_argument("--n", type=int, default=0, help="override units per GPU (msm: total terms); implies a single workload")
    ap.add_argument("--window", type=int, default=0, help="fixed-base / Pippenger window bits override")
    ap.add_argument("--sets", type=int, default=4, help="independently seeded input sets per workload, one per step in turn (default 4)")
    ap.add_argument("--check", action="store_true", help="(default) verify the last step of every workload against the oracle")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--sync-calls", action="store_true", help="fixed / variable base: one synchronous call per step instead of the queued calls")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="default run: skip the PCIe-inclusive host-pointer MSM (N = 1) / the single-process group MSM (N > 1)")
    ap.add_argument("--cpu-plumbing", action="store_true",
                    help="BASELINE configs[0] alone: 1024 k256 generator multiplications on the CPU reference path (no GPU needed)")
    ap.add_argument("--group-msm-child", type=int, default=0, help=argparse.SUPPRESS)     # internal: see group_msm_in_child
    args = ap.parse_args()
    if args.group_msm_child:
        print(json.dumps(group_msm_child_main(args)), flush=True)
        return None
    if args.cpu_plumbing:
        rec = cpu_plumbing_record()
        print(json.dumps(rec), flush=True)
        return rec

    b = Bench(args)
    top = args.only or "fixed_k256"
    cpu_leg = not args.no_cpu_baseline and b.world == 1          # the CPU leg is timed at N = 1 only
    single = bool(args.only or args.n or args.window)
    rec = b.run(top, cpu_leg)
    subs = [] if single else [x for x in DEFAULT_SUBS if b.world == 1 or x != "msm_k256_2p21"]
    full = {}
    for name in subs:
        # (the CPU leg beside BASELINE's own configs and the signature callers; the uniform-schedule twins share their oracle functions)
        r = b.run(name, cpu_leg and name not in ("var_p256_ct", "fixed_k256_ct"))
        if r is not None:
            full[name] = r
    lanes = {}
    if not single and not args.no_extras:       # the MSM sizes again with several MSMs in flight (top-level keys only)
        for name in (("msm_k256_lanes", "msm_k256_2p21_lanes", "msm_k256_2p21_sharded_lanes") if b.world == 1 else ("msm_k256_sharded_lanes",)):
            r = b.run(name, False)
            if r is not None:
                lanes[name] = r
    tiers = None
    if not single and b.world == 1 and not args.no_extras and rec is not None:
        # The library's DEFAULT table policy grows the generator table with use (16 -> 22 -> 26 bits after 2^26 / 2^29 multiplications on
        # the device); the headline asks for the widest up front.  What the same batch costs on the two narrower tiers:
        tiers = {"26": float("%.4g" % rec["ms_per_step"]), "check": True}
        for w in (16, 22):
            args.window = w
            r = b.run("fixed_k256", False)
            tiers[str(w)] = float("%.4g" % r["ms_per_step"])
            tiers["check"] = tiers["check"] and bool(r["check_vs_oracle"] or args.no_check)
        args.window = 0
        b.eng.set_base_window(b.ecgpu.CURVE_IDS["k256"], 0)
    e2e = e2e_fx = group = None
    if not single and not args.no_extras:
        if b.world == 1:
            e2e = b.e2e_msm()
            e2e_fx = b.e2e_fixed()
        else:
            if b.rank == 0:
                # in a child process with a deadline: the library's own RCCL communicator (ncclCommInitAll over all GPUs from
                # one process) meets this hardware for the first time here, and a hang must not cost the line above
                group = group_msm_in_child(args, b.world)
            # the other ranks wait on the HOST (gloo) while the child has the GPUs: an RCCL barrier here would leave a spinning collective
            # kernel on every other GPU for as long as the child runs (and inside the nccl group's watchdog timeout)
            b.dist.barrier()
            b.fence()
    if rec is not None and single:
        print(json.dumps(rec), flush=True)                       # profiling / sweep runs: the full record of one workload
    elif rec is not None:
        # ---- the driver's line: every BASELINE config in one record below 8 KB (the driver keeps the last 8 KB) ----
        rf = rec["roofline"]
        rec["roofline"] = {k: rf[k] for k in ("bound", "kernel", "kernel_ms", "kernel_ms_min", "workload_frac_8d", "achieved", "peak", "unit", "frac", "traffic", "mad_frac",
                                              "frac_cycles_pmc", "peak_probe", "frac_vs_probe", "clock_ghz_kernel", "algorithmic_speedup")}
        rec["roofline"]["hbm_algorithmic_gbps"] = rf["hbm"]["achieved"]
        rec["roofline"]["source"] = "profiles/roofline_consts.json"
        rec.pop("calls", None)
        if "cpu_baseline" in rec:
            cb = rec["cpu_baseline"]
            cb["sample"] = "%d threads x %d units of the seeded workload, ~%.1f s wall, after a 1-thread pilot" % (
                cb["cores"], int(round(cb["value"] * cb["wall_s"] / cb["cores"])), cb["wall_s"])
        for name in ("var_p256", "msm_k256", "var_p384"):        # BASELINE's other metrics as top-level keys
            if name in full:
                r = full[name]
                rec[name + "_value"] = r["value"]
                rec[name + "_ms_per_step"] = r["ms_per_step"]
                rec[name + "_frac"] = r["roofline"]["frac"]                       # the dominant kernel's executed work / roof
                rec[name + "_workload_frac"] = r["roofline"]["workload_frac_8d"]  # SURVEY 8d's numerator over the WHOLE step
                rec[name + "_check"] = r["check_vs_oracle"]
        if tiers:
            rec["fixed_k256_tier_ms"] = tiers       # ms per 2^20 scalars by table width (bits): the default (adaptive) policy's three tiers
        if e2e_fx:
            rec["fixed_k256_e2e_ms"] = e2e_fx["e2e_ms"]
            rec["fixed_k256_e2e_check"] = e2e_fx["e2e_check"]
        if "msm_k256" in full and e2e:
            full["msm_k256"].update(e2e)
            rec["msm_k256_e2e_ms"] = e2e["e2e_ms"]
        if "msm_k256" in full and "msm_k256_2p21" in full:
            share = full["msm_k256"]["ms_per_step"] / (8.0 * full["msm_k256_2p21"]["ms_per_step"])
            full["msm_k256_2p21"]["share_of_2p24_rate"] = share   # per-term rate of a 2^21-term share / per-term rate at 2^24
            full["msm_k256_2p21"]["projected_8gpu_efficiency"] = share
            rec["msm_k256_2p21_ms"] = full["msm_k256_2p21"]["ms_per_step"]
        for name, r in lanes.items():                            # throughput with several MSMs in flight per GPU (ecgpu_set_msm_lanes)
            rec[name + "_ms"] = r["ms_per_step"]
            rec[name + "_check"] = r["check_vs_oracle"]
        if "msm_k256_lanes" in lanes:                            # per-term rate of a 2^21-term share over the 2^24 rate, both with MSMs in flight
            for nm in ("msm_k256_2p21_lanes", "msm_k256_2p21_sharded_lanes"):
                if nm in lanes:
                    rec[nm + "_share"] = lanes["msm_k256_lanes"]["ms_per_step"] / (8.0 * lanes[nm]["ms_per_step"])
        rec["configs"] = {k: compact(v) for k, v in full.items()}
        if group:
            rec["configs"]["group_msm_k256"] = group
        if cpu_leg:
            c0 = cpu_plumbing_record()
            rec["configs"]["cpu_k256_1024"] = {k: c0[k] for k in ("metric", "value", "unit", "ms_per_step", "check_vs_model")}
        rec["cargo"] = cargo_probe()
        rec["notes"] = NOTES
        for k, v in list(rec.items()):                           # (sub-workload keys at six significant digits: the line stays below 8 KB)
            if isinstance(v, float) and k not in ("value", "ms_per_step"):
                rec[k] = float("%.6g" % v)
        rec["roofline"] = {k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in rec["roofline"].items()}
        line = json.dumps(rec, separators=(",", ":"))
        print(line, flush=True)
        if len(line) > 8000:
            print("bench.py: the record is %d bytes (> 8000)" % len(line), file=sys.stderr)
    b.close()
    if _SIGN_POOL is not None:
        _SIGN_POOL.shutdown()
    return rec


if __name__ == "__main__":
    main()
