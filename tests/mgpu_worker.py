"""One rank of the multi-GPU first-contact test (tests/test_gpu_multidevice.py launches WORLD of these under
torch.distributed.run, one process per GPU, backend "nccl" = RCCL over xGMI).

Every rank holds only its term shard in its own GPU's HBM, runs ecgpu_msm_parts_dev on it, the per-window partial sums
travel in ONE all_gather_into_tensor (sharded.RecordExchange), and ecgpu_msm_finish_dev combines them on every rank.  The
result must equal, byte for byte, on EVERY rank,
  * the single-GPU ecgpu_msm_dev of the whole problem (rank 0 computes it and broadcasts it), and
  * (sum_i k_i s_i mod n) G for the points P_i = s_i G the shards were built from — the exact dot product, whatever the
    term count (the check bench.py uses),
for an even split, an uneven one, a world with an EMPTY shard (fewer terms than ranks) and the plain / GLV plans.
Mirrors bench.py's N > 1 code path line for line (same Engine calls, same exchange object).
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist

    import oracle_lib
    ecgpu = importlib.import_module("elliptic-curves_amd")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("MGPU_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", rank))     # (one-GPU dry runs: every rank on device 0)
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    # as bench.py: gloo control plane, RCCL for the records only if a canary process and the group's first collective are
    # healthy (sharded.init_exchange); MGPU_REQUIRE_RCCL=1 (the first-contact test on real multi-GPU boxes) insists on it
    ex_info = ecgpu.init_exchange(torch, dist, local)
    if os.environ.get("MGPU_REQUIRE_RCCL") and ex_info.kind != "rccl":
        raise SystemExit("RCCL exchange not available: %s" % ex_info.reason)
    eng = ecgpu.Engine(local)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    oracle_lib.build()
    cases = [(ecgpu.K256, (1 << 16) + 3), (ecgpu.K256, 1 << 18), (ecgpu.P256, (1 << 15) + 1), (ecgpu.K256, world - 1), (ecgpu.P384, 5)]
    for cid, n_total in cases:
        L = ecgpu.FIELD_BYTES[cid]
        order = ecgpu.GROUP_ORDERS[cid]
        # the whole problem from one seed (every rank generates it identically and keeps its slice: the test needs the
        # single-GPU answer too; bench.py generates per-rank shards only)
        rng = np.random.default_rng(0xEC0003F7 + cid + n_total)
        k = oracle_lib.scalar_reduce(cid, rng.integers(0, 256, max(n_total, 1) * L, dtype=np.uint8))[: n_total * L]
        s = oracle_lib.scalar_reduce(cid, rng.integers(0, 256, max(n_total, 1) * L, dtype=np.uint8))[: n_total * L]
        lo, hi = ecgpu.shard_range(n_total, rank, world)
        n = hi - lo
        d_k = torch.from_numpy(k[lo * L: hi * L].copy()).to(device).reshape(n, L)
        d_s = torch.from_numpy(s[lo * L: hi * L].copy()).to(device).reshape(n, L)
        d_pts = torch.empty((max(n, 1), 2 * L), dtype=torch.uint8, device=device)
        torch.cuda.synchronize()
        if n:
            eng.mul_by_generator_dev(cid, d_s, n, d_pts, None)
        plan_terms = max(1, (n_total + world - 1) // world)
        ex = ecgpu.RecordExchange(torch, dist, eng.msm_parts_bytes(cid, plan_terms), device, group=ex_info.group)
        d_out = torch.zeros((1, 2 * L), dtype=torch.uint8, device=device)
        d_inf = torch.zeros((16,), dtype=torch.uint8, device=device)
        eng.msm_parts_dev(cid, d_k if n else None, d_pts if n else None, None, n, plan_terms, ex.mine)
        eng.msm_finish_dev(cid, ex.gather(), world, plan_terms, d_out, d_inf)
        torch.cuda.synchronize()
        got = bytes(d_out.cpu().numpy().reshape(-1)) + bytes([int(d_inf[0].item())])
        # (a) the exact dot product
        dot = sum(int.from_bytes(k[i * L:(i + 1) * L].tobytes(), "big") * int.from_bytes(s[i * L:(i + 1) * L].tobytes(), "big")
                  for i in range(n_total)) % order
        w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer(dot.to_bytes(L, "big"), np.uint8))
        assert got == bytes(w) + bytes([int(wf[0])]), "rank %d: sharded MSM != (sum k s) G for curve %d, n %d" % (rank, cid, n_total)
        # (b) the single-GPU pipeline on rank 0, broadcast
        ref = torch.zeros((2 * L + 16,), dtype=torch.uint8, device=device)
        if rank == 0:
            all_k = torch.from_numpy(k.copy()).to(device).reshape(n_total, L)
            all_s = torch.from_numpy(s.copy()).to(device).reshape(n_total, L)
            all_p = torch.empty((max(n_total, 1), 2 * L), dtype=torch.uint8, device=device)
            o1 = torch.zeros((1, 2 * L), dtype=torch.uint8, device=device)
            f1 = torch.zeros((16,), dtype=torch.uint8, device=device)
            torch.cuda.synchronize()
            if n_total:
                eng.mul_by_generator_dev(cid, all_s, n_total, all_p, None)
            eng.lincomb_dev(cid, all_k if n_total else None, all_p if n_total else None, None, n_total, o1, f1)
            torch.cuda.synchronize()
            ref[: 2 * L] = o1.view(-1)
            ref[2 * L] = f1[0]
        ref_h = ref.cpu()                                    # (the default group is gloo: host tensors)
        dist.broadcast(ref_h, 0)
        ref = ref_h.to(device)
        assert got == bytes(ref[: 2 * L + 1].cpu().numpy()), "rank %d: sharded MSM != single-GPU MSM (curve %d, n %d)" % (rank, cid, n_total)
        # (c) every rank holds the same bytes (all-gather of the result records)
        mine = torch.frombuffer(bytearray(got + bytes(15 - (len(got) - 1) % 16)), dtype=torch.uint8).to(device)
        allr = torch.empty((world * mine.numel(),), dtype=torch.uint8)
        dist.all_gather_into_tensor(allr, mine.cpu())
        rows = allr.numpy().reshape(world, -1)
        assert all(bytes(rows[r]) == bytes(rows[0]) for r in range(world)), "ranks disagree"
    dist.barrier()
    if rank == 0:
        print("MGPU_WORKER_OK world=%d exchange=%s (%s)" % (world, ex_info.kind, ex_info.reason), flush=True)
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
