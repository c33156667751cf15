#!/usr/bin/env python3
"""bench.py — throughput of the scalar-mul hot path on MI355X, one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--check]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM when the timed region starts.  Workloads (BASELINE.json `configs`):

    fixed_k256  (default, configs[1])  k256 fixed-base, 2^20 random scalars per GPU      -> scalar-muls/s
    var_p256    (configs[2])           p256 variable-base (ECDH shape), 2^20 pairs/GPU   -> scalar-muls/s
    var_p384    (configs[4])           p384 variable-base, 2^20 pairs per GPU            -> scalar-muls/s
    msm_k256    (configs[3])           k256 MSM, 2^24 terms in total, sharded over GPUs  -> terms/s

Batch workloads shard embarrassingly (weak scaling, no data-path collective).  The MSM shards its
terms (strong scaling) and has one exchange step: an RCCL all-gather of one affine point per rank
followed by a device point sum (elliptic-curves_amd/sharded.py).

`roofline` prices the dominant kernel against the integer-VALU roof (the path is
neither HBM- nor MFMA-bound): achieved = algorithmic IMAD32 per unit x units per launch / kernel time,
peak = the v_mad_u64_u32 rate measured on this GPU by ecgpu_valu_probe.  The HBM view of the same
launch is reported alongside under "hbm".  `cpu_baseline` times the oracle (a C restatement of the
reference's own CPU algorithm, kind "port") on the host cores for a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
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
    "fixed_k256": dict(curve="k256", kind="fixed", n=1 << 20, metric="k256 fixed-base scalar-muls/sec", unit="scalar-muls/s",
                       imad_per_unit=812 * 136, bytes_per_unit=96, kernel="k_fixed_base<K256Params>", scaling="weak"),
    "var_p256": dict(curve="p256", kind="var", n=1 << 20, metric="p256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=4336 * 136, bytes_per_unit=160, kernel="k_var_base<P256Params>", scaling="weak"),
    "var_k256": dict(curve="k256", kind="var", n=1 << 20, metric="k256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=1984 * 136, bytes_per_unit=160, kernel="k_var_base<K256Params>", scaling="weak"),
    "msm_p256": dict(curve="p256", kind="msm", n=1 << 24, metric="p256 MSM terms/sec", unit="terms/s",
                     imad_per_unit=int(16.06 * 11 * 136), bytes_per_unit=96 + 16 * 64, kernel="k_msm_accumulate<P256Params>",
                     scaling="strong"),
    # batch ECDSA verification: u1 G + u2 Q per signature = fixed-base + variable-base + 1 addition
    "ecdsa_p256": dict(curve="p256", kind="ecdsa", n=1 << 20, metric="p256 ECDSA verifications/sec", unit="verifications/s",
                       imad_per_unit=(962 + 4336 + 14) * 136, bytes_per_unit=160, kernel="k_var_base<P256Params>", scaling="weak"),
    "var_p384": dict(curve="p384", kind="var", n=1 << 20, metric="p384 variable-base scalar-muls/sec", unit="scalar-muls/s",
                     imad_per_unit=6448 * 300, bytes_per_unit=240, kernel="k_var_base<P384Params>", scaling="weak"),
    "msm_k256": dict(curve="k256", kind="msm", n=1 << 24, metric="k256 MSM terms/sec", unit="terms/s",
                     # Pippenger, algorithmic: ceil(256/c) * (N + 2^c) mixed adds at 11 M, c = 16
                     imad_per_unit=int(16.06 * 11 * 136), bytes_per_unit=96 + 16 * 64, kernel="k_msm_accumulate<K256Params>",
                     scaling="strong"),
}
def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/r01/pmc_traffic_v10.json: rocprofv3
    --pmc FETCH_SIZE and WRITE_SIZE in separate runs of this same command).  PMC counters cannot be collected
    from inside the timed process, so the figure is the committed measurement, valid for the default sizes."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "pmc_traffic_v10.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None, None, None
    if not rec:
        return None, None, None
    return (rec["fetch_bytes"] + rec["write_bytes"], "profiles/r01/pmc_traffic_v10.json (%s)" % rec["workload"],
            rec.get("valu_busy"))


HBM_PEAK_GBPS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3 TB/s achievable)


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


def cpu_baseline(wl, cid, L, sample_scalars, sample_points, extra=None):
    """Oracle ("port" of the reference's CPU algorithm) on the host cores, bounded to ~10-20 s of CPU work."""
    import oracle_lib
    oracle_lib.build()
    cores = os.cpu_count() or 1
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
        else:
            oracle_lib.msm(cid, s, sample_points[lo * 2 * L: hi * 2 * L], vartime=True)

    avail = sample_scalars.size // L
    pilot = min(avail, 256 if kind != "fixed" else 1024)
    t0 = time.perf_counter()
    run(0, pilot)
    per_unit = (time.perf_counter() - t0) / pilot
    single = 1.0 / per_unit
    target_cpu_seconds = 12.0
    total = int(min(avail, max(cores * 64, target_cpu_seconds / per_unit)))
    chunk = max(1, total // cores)
    total = chunk * cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda i: run(i * chunk, (i + 1) * chunk), range(cores)))
    dt = time.perf_counter() - t0
    algo = {"fixed": "mul_by_generator (33/49-LUT basepoint table)", "var": "ProjectivePoint * Scalar (LUT + radix-16)",
            "msm": "lincomb_vartime (GLV + wNAF-5 Straus), per-thread chunks summed",
            "ecdsa": "verify_prehashed: s^-1, u1 G + u2 Q (mul_by_generator_and_mul_add_vartime), x mod n == r"}[kind]
    return {"value": total / dt, "unit": wl["unit"], "cores": cores, "kind": "port",
            "sample": "%d units of the same seeded workload, %s, oracle/ C restatement, %d threads" % (total, algo, cores),
            "single_thread_value": single}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fixed_k256", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override units per GPU (msm: total terms)")
    ap.add_argument("--window", type=int, default=0, help="fixed-base / Pippenger window bits override")
    ap.add_argument("--check", action="store_true", help="verify a sample of the last step against the oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a gfx950 GPU; there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(device))

    ecgpu = importlib.import_module("elliptic-curves_amd")
    eng = ecgpu.Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    wl = WORKLOADS[args.workload]
    cid = ecgpu.CURVE_IDS[wl["curve"]]
    L = ecgpu.FIELD_BYTES[cid]
    kind = wl["kind"]
    n_total = args.n or wl["n"]
    if kind == "msm":
        lo, hi = ecgpu.shard_range(n_total, rank, world)
        n = hi - lo
        if args.window:
            eng.set_msm_window(args.window)
    else:
        n = n_total
        if args.window and kind == "fixed":
            eng.set_base_window(cid, args.window)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    seed = 0xEC000000 + {"fixed_k256": 2, "var_p256": 3, "msm_k256": 4, "var_p384": 5, "var_k256": 6, "ecdsa_p256": 7, "msm_p256": 8}[args.workload] + 1000 * rank
    d_scal = device_random_scalars(torch, n, L, seed, device)
    d_pts = d_out = None
    if kind in ("var", "msm"):
        d_s2 = device_random_scalars(torch, n, L, seed + 50, device)
        d_pts = torch.empty((n, 2 * L), dtype=torch.uint8, device=device)
        torch.cuda.synchronize()
        eng.mul_by_generator_dev(cid, d_s2, n, d_pts, None)          # P_i = s_i * G (untimed setup)
        del d_s2
    d_r = d_s = d_ok = None
    if kind == "ecdsa":
        # valid signatures: 2^16 distinct (d, k, z) triples signed on the host from k*G computed here, tiled to n
        m = min(n, 1 << 16)
        d_d = device_random_scalars(torch, m, L, seed + 50, device)
        d_k = device_random_scalars(torch, m, L, seed + 51, device)
        d_Q = torch.empty((m, 2 * L), dtype=torch.uint8, device=device)
        d_R = torch.empty((m, 2 * L), dtype=torch.uint8, device=device)
        torch.cuda.synchronize()
        eng.mul_by_generator_dev(cid, d_d, m, d_Q, None)
        eng.mul_by_generator_dev(cid, d_k, m, d_R, None)
        torch.cuda.synchronize()
        n_order = ecgpu.GROUP_ORDERS[cid]
        dh, kh, zh, rx = (t.cpu().numpy() for t in (d_d, d_k, d_scal[:m], d_R[:, :L].contiguous()))
        rb, sb = bytearray(), bytearray()
        for i in range(m):
            di, ki = int.from_bytes(dh[i].tobytes(), "big"), int.from_bytes(kh[i].tobytes(), "big") or 1
            zi, ri = int.from_bytes(zh[i].tobytes(), "big"), int.from_bytes(rx[i].tobytes(), "big") % n_order
            si = pow(ki, -1, n_order) * (zi + ri * di) % n_order
            rb += ri.to_bytes(L, "big"); sb += si.to_bytes(L, "big")
        reps = (n + m - 1) // m
        d_r = torch.frombuffer(rb, dtype=torch.uint8).reshape(m, L).to(device).repeat(reps, 1)[:n].contiguous()
        d_s = torch.frombuffer(sb, dtype=torch.uint8).reshape(m, L).to(device).repeat(reps, 1)[:n].contiguous()
        d_scal = d_scal[:m].repeat(reps, 1)[:n].contiguous()
        d_pts = d_Q.repeat(reps, 1)[:n].contiguous()
        d_ok = torch.zeros((n + 16,), dtype=torch.uint8, device=device)
        del d_d, d_k, d_R, d_Q
    n_out = 1 if kind == "msm" else n
    d_out = torch.empty((n_out, 2 * L), dtype=torch.uint8, device=device)
    d_inf = torch.empty((max(n_out, 16),), dtype=torch.uint8, device=device)
    exchange = ecgpu.TensorExchange(torch, dist, L, device) if kind == "msm" and world > 1 else None
    torch.cuda.synchronize()     # inputs were written on torch's stream; the engine works on its own (non-blocking) stream

    main_ms = []

    def step():
        if kind == "fixed":
            eng.mul_by_generator_dev(cid, d_scal, n, d_out, d_inf)
        elif kind == "var":
            eng.mul_dev(cid, d_scal, d_pts, None, n, d_out, d_inf)
        elif kind == "ecdsa":
            eng.ecdsa_verify_dev(cid, d_scal, d_r, d_s, d_pts, n, False, d_ok)
        else:
            eng.lincomb_dev(cid, d_scal, d_pts, None, n, d_out, d_inf)
        main_ms.append(eng.last_timing("accumulate" if kind == "msm" else "main") or 0.0)
        if exchange is not None:                                       # the one exchange step: all-gather + EC sum
            exchange.combine(lambda pts, flags, w, oxy, oinf: eng.point_sum_dev(cid, pts, flags, w, oxy, oinf), d_out, d_inf)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    main_ms.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    units_per_step = n_total if kind == "msm" else n * world
    value = units_per_step * args.steps / elapsed

    result = None
    if rank == 0:
        kernel_ms = float(np.mean(main_ms)) if main_ms else None
        peak = eng.valu_probe(0)                                       # v_mad_u64_u32 / s on this GPU
        units_per_launch = n
        achieved = wl["imad_per_unit"] * units_per_launch / (kernel_ms * 1e-3) if kernel_ms else None
        hbm_gbps = wl["bytes_per_unit"] * units_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
        default_size = (args.n == 0 and not args.window and world == 1)
        traffic, traffic_src, valu_busy = pmc_traffic(wl["kernel"]) if default_size else (None, None, None)
        result = {
            "metric": wl["metric"], "value": value, "unit": wl["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": wl["scaling"], "vs_baseline": None, "dtype": "u32 limbs (v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": args.workload, "curve": wl["curve"], "units_per_gpu": n, "units_total": units_per_step,
                       "window_bits": args.window or "default", "parallelism": "shard%d" % world},
            "roofline": {"bound": "valu-int", "kernel": wl["kernel"], "kernel_ms": kernel_ms,
                         "achieved": achieved / 1e12 if achieved else None, "peak": peak / 1e12, "unit": "TIMAD32/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "valu_busy_pmc": valu_busy,
                         "algorithmic_imad32_per_unit": wl["imad_per_unit"], "units_per_launch": units_per_launch,
                         "peak_source": "ecgpu_valu_probe(v_mad_u64_u32) measured in this run",
                         "frac_note": "numerator = the reference algorithm's IMAD32 count (SURVEY.md 8d); above 1 means the "
                                      "GPU algorithm does less arithmetic per unit; utilisation of the VALU issue roof is "
                                      "valu_busy_pmc (DESIGN.md 6)",
                         "hbm": {"achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": hbm_gbps / HBM_PEAK_GBPS if hbm_gbps else None,
                                 "algorithmic_bytes_per_unit": wl["bytes_per_unit"]}},
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed at N = 1 only
            ns = min(n, 1 << 17 if kind == "fixed" else (1 << 14 if kind == "var" else 1 << 15))
            s_host = d_scal[:ns].cpu().numpy().reshape(-1)
            p_host = d_pts[:ns].cpu().numpy().reshape(-1) if d_pts is not None else None
            extra = (d_r[:ns].cpu().numpy().reshape(-1), d_s[:ns].cpu().numpy().reshape(-1)) if kind == "ecdsa" else None
            result["cpu_baseline"] = cpu_baseline(wl, cid, L, s_host, p_host, extra)
        if args.check:
            import oracle_lib
            oracle_lib.build()
            if kind == "msm":
                if n_total <= (1 << 16) and world == 1:
                    w, wf = oracle_lib.msm(cid, d_scal.cpu().numpy().reshape(-1), d_pts.cpu().numpy().reshape(-1), vartime=True)
                    ok = bytes(w) == bytes(d_out[0].cpu().numpy()) and wf == int(d_inf[0].item())
                else:
                    ok = None
            elif kind == "ecdsa":
                m = min(n, 256)
                w = oracle_lib.ecdsa_verify(cid, d_scal[:m].cpu().numpy().reshape(-1), d_r[:m].cpu().numpy().reshape(-1),
                                            d_s[:m].cpu().numpy().reshape(-1), d_pts[:m].cpu().numpy().reshape(-1))
                ok = bool(w.all()) and bool(d_ok[:n].all().item())       # every synthetic signature is valid
            else:
                m = min(n, 256)
                got = d_out[:m].cpu().numpy().reshape(-1)
                sh = d_scal[:m].cpu().numpy().reshape(-1)
                if kind == "fixed":
                    w, _ = oracle_lib.batch_mul_base(cid, sh)
                else:
                    w, _ = oracle_lib.batch_mul(cid, sh, d_pts[:m].cpu().numpy().reshape(-1))
                ok = bytes(w) == bytes(got)
            result["check_vs_oracle"] = ok
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    return result


if __name__ == "__main__":
    main()
