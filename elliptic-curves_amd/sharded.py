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

First contact with RCCL must not be able to lose a run (bench.py --gpus N is launched ONCE by the driver on hardware this
code has never seen): `init_exchange` below brings the process group up on gloo (the control plane: barriers, the agreement
on the exchange path), tries RCCL in a throw-away CHILD process per rank first (`python sharded.py --nccl-probe`: its own
rendezvous port, one all-gather, a deadline) and only when every rank's canary came back healthy creates the nccl group the
data path uses.  A canary that crashes, reports wrong bytes or hangs past its deadline is killed, and the job runs its
exchange through gloo (the records are tens of KiB) with the reason on record — a measured scaling curve with a slower
exchange instead of a crash.
"""
import os
import subprocess
import sys
import time

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

    def gather(self, consumer_on_current_stream=False):
        """All-gathers `mine` into `all` and returns it.  consumer_on_current_stream: what reads `all` next is queued on torch's current
        stream (an Engine after set_stream(torch.cuda.current_stream()), as bench.py's): the collective is ordered before it on the
        device and the host does not wait; otherwise (an Engine on its own stream) the host waits for the gathered bytes."""
        if self.all.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # dry runs on a one-GPU box (bench.py ECGPU_BENCH_BACKEND=gloo): the records travel through the host
            self.torch.cuda.current_stream(self.all.device).synchronize()
            host = self.torch.empty_like(self.all, device="cpu")
            self.dist.all_gather_into_tensor(host, self.mine.cpu(), group=self.group)
            self.all.copy_(host)
        else:
            self.dist.all_gather_into_tensor(self.all, self.mine, group=self.group)
        if self.all.is_cuda and not consumer_on_current_stream:
            # a consumer that enqueues on a stream of its own (an Engine without set_stream) is not ordered after the
            # collective: wait for the gathered bytes
            self.torch.cuda.current_stream(self.all.device).synchronize()
        return self.all


class LocalRecord:
    """RecordExchange's shape for ONE rank: the record is its own gathered form (`all` is `mine`), gather() moves nothing.  What
    bench.py uses to time a single GPU's share of a sharded MSM in the sharded form (parts / join / finish) without a process group."""

    def __init__(self, torch, nbytes, device):
        pad = (nbytes + 255) // 256 * 256
        self.nbytes, self.world = nbytes, 1
        self.mine = torch.zeros((pad,), dtype=torch.uint8, device=device)[:nbytes]
        self.all = self.mine

    def gather(self, consumer_on_current_stream=False):
        return self.all


# ---- bringing the exchange up: gloo control plane, RCCL canary, fallback ------------------------------------------------------

PROBE_BYTES = 64 << 10          # one parts record of a k256 MSM is 41 KiB


def _nccl_probe_main():
    """The canary (child process): RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT from the environment, one
    all_gather_into_tensor over nccl (= RCCL on ROCm) on this rank's GPU, contents checked.  Exit status 0 = healthy."""
    from datetime import timedelta

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("ECGPU_PROBE_DEVICE", os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev, timeout=timedelta(seconds=int(os.environ.get("ECGPU_PROBE_TIMEOUT", "60"))))
    mine = torch.full((PROBE_BYTES,), rank + 1, dtype=torch.uint8, device=dev)
    every = torch.zeros((world * PROBE_BYTES,), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(every, mine)
    torch.cuda.synchronize()
    got = every.view(world, PROBE_BYTES)[:, ::4096].cpu().numpy()
    ok = all((got[r] == r + 1).all() for r in range(world))
    dist.destroy_process_group()
    print("NCCL_PROBE_OK" if ok else "NCCL_PROBE_BAD_BYTES", flush=True)
    return 0 if ok else 3


def run_nccl_probe(local_device, port_offset=1, timeout=None):
    """Runs the canary for this rank; -> (healthy: bool, reason: str).  Never raises, never outlives `timeout` seconds."""
    timeout = float(timeout if timeout is not None else os.environ.get("ECGPU_NCCL_PROBE_TIMEOUT", "150"))
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["ECGPU_PROBE_DEVICE"] = str(local_device)
    env["ECGPU_PROBE_TIMEOUT"] = str(int(max(20, timeout - 30)))
    # The canary makes a rendezvous of its own on MASTER_PORT + port_offset, where rank 0's child must HOST the store.  Under
    # `python -m torch.distributed.run` — the driver's launch — the workers inherit TORCHELASTIC_USE_AGENT_STORE=True, which makes
    # every rank a CLIENT of the agent's store: with it left in place nobody listens on the canary's port, every canary times out in
    # its rendezvous ("The client socket has timed out ... (127.0.0.1, 29542)", the reason on record in profiles/r04 and r05's
    # N-rank dry runs) and a healthy 8-GPU node would have been sent to the gloo fallback without RCCL ever being tried.
    for k in [k for k in env if k.startswith("TORCHELASTIC_")] + ["TORCH_NCCL_ASYNC_ERROR_HANDLING"]:
        env.pop(k, None)
    if os.environ.get("ECGPU_NCCL_PROBE_FAIL"):            # fault injection for the dry runs / tests: crash | hang
        mode = os.environ["ECGPU_NCCL_PROBE_FAIL"]
        cmd = [sys.executable, "-c", "import time, sys; time.sleep(3600)" if mode == "hang" else "import sys; sys.exit('injected canary failure')"]
    else:
        cmd = [sys.executable, os.path.abspath(__file__), "--nccl-probe"]
    t0 = time.time()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except OSError as e:
        return False, "canary could not be started: %s" % e
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        try:
            p.communicate(timeout=10)
        except Exception:
            pass
        return False, "RCCL canary still running after %.0f s (killed)" % timeout
    if p.returncode == 0 and "NCCL_PROBE_OK" in out:
        return True, "RCCL canary healthy in %.1f s" % (time.time() - t0)
    tail = [ln for ln in (err or out or "").strip().splitlines() if ln.strip()]
    return False, "RCCL canary exit %s: %s" % (p.returncode, (tail[-1] if tail else "no output")[:300])


class Exchange:
    """What init_exchange returns: `group` for RecordExchange / TensorExchange (None = the default gloo group), `kind`
    ("rccl" | "gloo-fallback" | "gloo-forced"), `reason` (one line for the bench record)."""

    def __init__(self, kind, reason, group):
        self.kind, self.reason, self.group = kind, reason, group


def init_exchange(torch, dist, local_device, prefer="nccl", probe=True):
    """Initialises torch.distributed for a one-process-per-GPU job so that it cannot be lost to the collective library:
    default group = gloo (always); the data-path group is nccl only if every rank's canary was healthy AND the group's own
    first collective completes; otherwise the exchange stays on gloo.  Call once per process, after torch.cuda.set_device."""
    from datetime import timedelta

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group("gloo", timeout=timedelta(seconds=600))
    if prefer != "nccl":
        return Exchange("gloo-forced", "exchange backend forced to %s" % prefer, None)
    ok, reason = run_nccl_probe(local_device) if probe else (True, "canary skipped")
    vote = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)                     # gloo: every rank learns whether ALL canaries were healthy
    if int(vote.item()) != 1:
        reasons = [None] * dist.get_world_size()
        dist.all_gather_object(reasons, reason)
        bad = [(r, x) for r, x in enumerate(reasons) if "healthy" not in x and "skipped" not in x]
        return Exchange("gloo-fallback", "rank %d: %s" % bad[0] if bad else reason, None)
    try:
        group = dist.new_group(backend="nccl", timeout=timedelta(seconds=180), device_id=torch.device("cuda", local_device))
        t = torch.full((1024,), dist.get_rank() + 1, dtype=torch.uint8, device="cuda:%d" % local_device)
        every = torch.zeros((dist.get_world_size() * 1024,), dtype=torch.uint8, device="cuda:%d" % local_device)
        dist.all_gather_into_tensor(every, t, group=group)
        torch.cuda.synchronize()
        good = bool((every.view(-1, 1024)[:, 0].cpu() == torch.arange(1, dist.get_world_size() + 1, dtype=torch.uint8)).all())
        if not good:
            reason = "nccl group in the main process returned wrong bytes from its first all-gather"
    except Exception as e:                                           # raised errors only: a hang here is what the canary is for
        good, group, reason = False, None, "nccl group in the main process failed: %s" % str(e).splitlines()[0][:300]
    vote = torch.tensor([1 if good else 0], dtype=torch.int32)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if int(vote.item()) != 1:
        if group is not None:                                        # not to be used: do not leave it alive beside the gloo fallback
            try:
                dist.destroy_process_group(group)
            except Exception:
                pass
        return Exchange("gloo-fallback", reason if not good else "another rank's nccl group failed", None)
    return Exchange("rccl", reason, group)


if __name__ == "__main__":
    if "--nccl-probe" in sys.argv:
        sys.exit(_nccl_probe_main())
