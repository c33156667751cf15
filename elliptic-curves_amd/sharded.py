"""Sharded MSM across the GPUs of one node, one process per GPU over torch.distributed.

sum_i k_i P_i = sum_ranks ( sum_{i in shard(rank)} k_i P_i ): terms are partitioned into contiguous
slices, every rank runs the local Pippenger pipeline on its slice, and the only exchange step is
an all-gather of one fixed-size record per rank followed by a combining step on every rank.
RCCL's built-in reductions cannot add curve points, so "all-reduce of partial sums" is realised as
all-gather + a device combine.  Two record kinds:
  * RecordExchange with ecgpu_msm_parts_dev / ecgpu_msm_finish_dev (bench.py): the record is the GPU's per-window
    partial sums (41 KiB for k256) and the combine is the window sums over all ranks + ONE Horner chain — the serial
    tail of the bucket method runs once instead of once per rank plus a point sum;
  * lincomb_sharded / TensorExchange: the record is one affine point per rank (2L + 1 bytes), the combine a point sum
    (ecgpu_point_sum) — what a caller without the split entry points does.
Either way the payload is far below a megabyte: the step is latency- not bandwidth-bound on xGMI.

The compute steps are injected, so the same host logic is exercised on CPU by the gloo tests (with the
oracle standing in for the GPU) and on the GPU box by bench.py / the -m gpu tests (with Engine methods).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of n terms for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def lincomb_sharded(L, local_lincomb, point_sum, scalars, points_xy, points_inf=None, *, dist=None, group=None,
                    device=None, pre_sharded=False):
    """Whole-job lincomb over all ranks.

    L                field bytes of the curve (32 / 48)
    local_lincomb    f(scalars, points_xy, points_inf) -> (xy uint8[2L], inf int)   this rank's MSM
    point_sum        f(points_xy uint8[w*2L], points_inf uint8[w]) -> (xy, inf)      EC sum of the partials
    scalars/points   the FULL problem (sliced here) or, with pre_sharded=True, this rank's slice
    dist             torch.distributed (initialised) or None for a single process
    device           torch device for the exchanged tensor ("cuda:k" for RCCL, "cpu" for gloo)
    Returns (xy, inf) identical on every rank.
    """
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
    p = np.ascontiguousarray(points_xy, dtype=np.uint8).reshape(-1)
    pi = None if points_inf is None else np.ascontiguousarray(points_inf, dtype=np.uint8).reshape(-1)
    if not pre_sharded:
        n = s.size // L
        lo, hi = shard_range(n, rank, world)
        s, p = s[lo * L: hi * L], p[lo * 2 * L: hi * 2 * L]
        if pi is not None:
            pi = pi[lo:hi]
    xy, inf = local_lincomb(s, p, pi)
    if world == 1:
        return np.asarray(xy, dtype=np.uint8), int(inf)
    import torch

    rec = np.zeros(2 * L + 16, np.uint8)           # x || y || flag, padded to a 16-byte multiple
    rec[: 2 * L] = np.asarray(xy, dtype=np.uint8)
    rec[2 * L] = int(inf)
    mine = torch.from_numpy(rec).to(device or "cpu")
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    allrec = torch.stack(gathered).cpu().numpy()
    pts = np.ascontiguousarray(allrec[:, : 2 * L]).reshape(-1)
    flags = np.ascontiguousarray(allrec[:, 2 * L])
    out_xy, out_inf = point_sum(pts, flags)
    return np.asarray(out_xy, dtype=np.uint8), int(out_inf)


class TensorExchange:
    """The exchange step of the sharded MSM on torch tensors that stay on their device (bench.py: HBM + RCCL; the
    gloo test: CPU tensors).  One record of 2L + 16 bytes (x || y || flag, padded to 16 bytes) per rank."""

    def __init__(self, torch, dist, L, device):
        self.torch, self.dist, self.L = torch, dist, L
        self.world = dist.get_world_size()
        self.rec = torch.zeros((2 * L + 16,), dtype=torch.uint8, device=device)
        self.all = torch.zeros((self.world, 2 * L + 16), dtype=torch.uint8, device=device)

    def combine(self, point_sum, out_xy, out_inf):
        """out_xy[0] (2L bytes) / out_inf[0] hold this rank's partial sum; on return they hold the sum over all
        ranks.  point_sum(points[world, 2L], flags[world], world, out_xy, out_inf) adds the gathered points."""
        L = self.L
        self.rec[: 2 * L] = out_xy.view(-1)[: 2 * L]
        self.rec[2 * L] = out_inf.view(-1)[0]
        self.dist.all_gather_into_tensor(self.all.view(-1), self.rec)
        pts = self.all[:, : 2 * L].contiguous()
        flags = self.all[:, 2 * L].contiguous()
        if pts.is_cuda:
            # the collective and the slicing run on torch's streams; the engine enqueues on its own stream, which is not
            # ordered after them (include/ecgpu.h, device-pointer entry points): wait for the gathered bytes first
            self.torch.cuda.current_stream(pts.device).synchronize()
        point_sum(pts, flags, self.world, out_xy, out_inf)


class RecordExchange:
    """The exchange step of a sharded computation on fixed-size opaque records that stay on their device:
    `mine` (nbytes uint8) is filled by the local half, gather() all-gathers it into `all` (world * nbytes, rank-major), and
    the combining half reads `all`.  With CUDA tensors the collective is RCCL over xGMI; with CPU tensors (the gloo
    tests) the same logic runs on the host."""

    def __init__(self, torch, dist, nbytes, device, group=None):
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        pad = (nbytes + 255) // 256 * 256
        self.nbytes = nbytes
        self.mine = torch.zeros((pad,), dtype=torch.uint8, device=device)[:nbytes]
        self.all = torch.zeros((self.world * nbytes,), dtype=torch.uint8, device=device)

    def gather(self):
        if self.all.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # dry runs on a one-GPU box (bench.py ECGPU_BENCH_BACKEND=gloo): the records travel through the host
            self.torch.cuda.current_stream(self.all.device).synchronize()
            host = self.torch.empty_like(self.all, device="cpu")
            self.dist.all_gather_into_tensor(host, self.mine.cpu(), group=self.group)
            self.all.copy_(host)
        else:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
        if self.all.is_cuda:
            # a consumer that enqueues on a stream of its own (an Engine without set_stream) is not ordered after the
            # collective: wait for the gathered bytes
            self.torch.cuda.current_stream(self.all.device).synchronize()
        return self.all
