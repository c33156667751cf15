"""world_size-2 gloo test of the sharded-MSM host logic (elliptic-curves_amd/sharded.py).  The local
compute is injected: here the oracle stands in for the GPU so that the slice / all-gather / combine
logic is covered on CPU; on the GPU box the same function runs with Engine methods."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_lib, pyec
ecgpu = importlib.import_module("elliptic-curves_amd")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=int(sys.argv[2]))
curve = int(sys.argv[3]); n = int(sys.argv[4])
L = oracle_lib.FIELD_BYTES[curve]
rng = np.random.default_rng(1234)
scal = oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8))
pts, _ = oracle_lib.batch_mul_base(curve, oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8)))
inf = np.zeros(n, np.uint8)
if n > 3:
    inf[2] = 1; pts[2 * 2 * L: 3 * 2 * L] = 0
local = lambda s, p, pi: oracle_lib.msm(curve, s, p, pi)
def psum(p, f):
    # sum of the partial points = lincomb with all scalars 1
    ones = np.tile(np.array([0] * (L - 1) + [1], np.uint8), f.size)
    return oracle_lib.msm(curve, ones, p, f)
xy, i = ecgpu.lincomb_sharded(L, local, psum, scal, pts, inf, dist=dist, device="cpu")
want, wi = oracle_lib.msm(curve, scal, pts, inf)
assert bytes(xy) == bytes(want) and i == wi, "rank %s mismatch" % sys.argv[1]
lo, hi = ecgpu.shard_range(n, dist.get_rank(), dist.get_world_size())
print("rank", dist.get_rank(), "ok", lo, hi)
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    ecgpu = importlib.import_module("elliptic-curves_amd")
    for n in (0, 1, 2, 7, 8, 9, 1 << 20, (1 << 24) + 5):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = ecgpu.shard_range(n, r, world)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == n and max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("curve,n", [(0, 37), (1, 20), (0, 1)])
def test_lincomb_sharded_world2_gloo(oracle, tmp_path, curve, n):
    port = 29500 + (os.getpid() + curve * 7 + n) % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(curve), str(n)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert " ok " in o


WORKER_TENSOR = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_lib
ecgpu = importlib.import_module("elliptic-curves_amd")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=int(sys.argv[2]))
curve = int(sys.argv[3]); n = int(sys.argv[4])
L = oracle_lib.FIELD_BYTES[curve]
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(4321)
scal = oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8))
pts, _ = oracle_lib.batch_mul_base(curve, oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8)))
lo, hi = ecgpu.shard_range(n, rank, world)                      # bench.py's term partition
xy, inf = oracle_lib.msm(curve, scal[lo * L: hi * L], pts[lo * 2 * L: hi * 2 * L])
out_xy = torch.from_numpy(np.asarray(xy, np.uint8).copy()).reshape(1, 2 * L)
out_inf = torch.zeros(16, dtype=torch.uint8); out_inf[0] = int(inf)
def point_sum(p, f, w, oxy, oinf):                             # stands in for Engine.point_sum_dev
    ones = np.tile(np.array([0] * (L - 1) + [1], np.uint8), w)
    sxy, sinf = oracle_lib.msm(curve, ones, p.numpy().reshape(-1), f.numpy())
    oxy.view(-1)[: 2 * L] = torch.from_numpy(np.asarray(sxy, np.uint8).copy()); oinf[0] = int(sinf)
ex = ecgpu.TensorExchange(torch, dist, L, "cpu")
for _ in range(2):                                             # the buffers are reused every step
    out_xy.view(-1)[: 2 * L] = torch.from_numpy(np.asarray(xy, np.uint8).copy()); out_inf[0] = int(inf)
    ex.combine(point_sum, out_xy, out_inf)
want, wi = oracle_lib.msm(curve, scal, pts)
assert bytes(out_xy.numpy().reshape(-1)) == bytes(want) and int(out_inf[0]) == wi, "rank %d mismatch" % rank
print("rank", rank, "ok")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("curve,n", [(0, 33), (2, 9)])
def test_tensor_exchange_world2_gloo(oracle, tmp_path, curve, n):
    """bench.py's exchange step (all_gather_into_tensor of one record per rank + point sum) on CPU tensors."""
    port = 31500 + (os.getpid() + curve * 11 + n) % 2000
    script = tmp_path / "worker_tensor.py"
    script.write_text(WORKER_TENSOR.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(curve), str(n)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert " ok" in o


WORKER_RECORD = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_lib
ecgpu = importlib.import_module("elliptic-curves_amd")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=int(sys.argv[2]))
curve = int(sys.argv[3]); n = int(sys.argv[4])
L = oracle_lib.FIELD_BYTES[curve]
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(9876)
scal = oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8))
pts, _ = oracle_lib.batch_mul_base(curve, oracle_lib.scalar_reduce(curve, rng.integers(0, 256, n * L, dtype=np.uint8)))
lo, hi = ecgpu.shard_range(n, rank, world)                      # bench.py's term partition
# the record stands in for ecgpu_msm_parts_dev's output: here the shard's sum as x || y || flag
nbytes = 2 * L + 1
ex = ecgpu.RecordExchange(torch, dist, nbytes, "cpu")
assert ex.mine.numel() == nbytes and ex.all.numel() == world * nbytes
for step in range(2):                                          # the buffers are reused every step
    xy, inf = oracle_lib.msm(curve, scal[lo * L: hi * L], pts[lo * 2 * L: hi * 2 * L])
    ex.mine[: 2 * L] = torch.from_numpy(np.asarray(xy, np.uint8).copy()); ex.mine[2 * L] = int(inf)
    rec = ex.gather().numpy().reshape(world, nbytes)           # stands in for ecgpu_msm_finish_dev's input
    ones = np.tile(np.array([0] * (L - 1) + [1], np.uint8), world)
    got, gi = oracle_lib.msm(curve, ones, np.ascontiguousarray(rec[:, : 2 * L]).reshape(-1), np.ascontiguousarray(rec[:, 2 * L]))
    want, wi = oracle_lib.msm(curve, scal, pts)
    assert bytes(got) == bytes(want) and gi == wi, "rank %d mismatch" % rank
# bench.py's throughput form (msm_k256_sharded_lanes): local half of step i, THEN exchange + combining half of step i - 1, on two
# records that take turns; gather(consumer_on_current_stream=True) as the engine on torch's stream uses it
exs = [ecgpu.RecordExchange(torch, dist, nbytes, "cpu") for _ in range(2)]
pend, results = [], []
def combine():
    b, j = pend.pop(0)
    rec = exs[b].gather(consumer_on_current_stream=True).numpy().reshape(world, nbytes)
    got, gi = oracle_lib.msm(curve, ones, np.ascontiguousarray(rec[:, : 2 * L]).reshape(-1), np.ascontiguousarray(rec[:, 2 * L]))
    results.append((j, bytes(got), gi))
for i in range(5):
    k_i = oracle_lib.scalar_reduce(curve, np.random.default_rng(100 + i).integers(0, 256, n * L, dtype=np.uint8))
    xy, inf = oracle_lib.msm(curve, k_i[lo * L: hi * L], pts[lo * 2 * L: hi * 2 * L])
    b = i % 2
    exs[b].mine[: 2 * L] = torch.from_numpy(np.asarray(xy, np.uint8).copy()); exs[b].mine[2 * L] = int(inf)
    pend.append((b, i))
    if len(pend) > 1:
        combine()
while pend:
    combine()
assert [j for j, _, _ in results] == list(range(5))
for j, got, gi in results:
    k_j = oracle_lib.scalar_reduce(curve, np.random.default_rng(100 + j).integers(0, 256, n * L, dtype=np.uint8))
    want, wi = oracle_lib.msm(curve, k_j, pts)
    assert got == bytes(want) and gi == wi, "rank %d, pipelined step %d" % (rank, j)
# one rank's form of the record (bench.py at N = 1): its own gathered form
lr = ecgpu.LocalRecord(torch, nbytes, "cpu")
lr.mine[:] = 7
assert lr.gather(consumer_on_current_stream=True).data_ptr() == lr.mine.data_ptr() and lr.world == 1 and int(lr.all[nbytes - 1]) == 7
print("rank", rank, "ok")
dist.destroy_process_group()
'''


@pytest.mark.parametrize("curve,n", [(0, 41), (1, 2)])
def test_record_exchange_world2_gloo(oracle, tmp_path, curve, n):
    """bench.py's sharded-MSM exchange (RecordExchange: all_gather_into_tensor of one opaque fixed-size record per rank) on
    CPU tensors; the records are the oracle's shard sums where the GPU path exchanges per-window partial sums."""
    port = 33500 + (os.getpid() + curve * 13 + n) % 2000
    script = tmp_path / "worker_record.py"
    script.write_text(WORKER_RECORD.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(curve), str(n)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert " ok" in o


WORKER_INIT = '''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
ecgpu = importlib.import_module("elliptic-curves_amd")
os.environ.update(RANK=sys.argv[1], WORLD_SIZE=sys.argv[2], LOCAL_RANK=sys.argv[1], MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
ex = ecgpu.init_exchange(torch, dist, int(sys.argv[1]), prefer=sys.argv[3])
rank, world = dist.get_rank(), dist.get_world_size()
assert ex.group is None and dist.get_backend() == "gloo", (ex.kind, ex.reason)
# the exchange object bench.py builds works on the path that was chosen
rx = ecgpu.RecordExchange(torch, dist, 48, "cpu", group=ex.group)
rx.mine[:] = rank + 1
got = rx.gather().numpy().reshape(world, 48)
assert all((got[r] == r + 1).all() for r in range(world))
print("rank", rank, "exchange", ex.kind, "|", ex.reason)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("prefer,fault,kind", [("nccl", "", "gloo-fallback"), ("nccl", "crash", "gloo-fallback"), ("nccl", "hang", "gloo-fallback"),
                                               ("gloo", "", "gloo-forced")])
def test_init_exchange_falls_back_to_gloo_world2(tmp_path, prefer, fault, kind):
    """sharded.init_exchange (what bench.py --gpus N and tests/mgpu_worker.py bring the job up with): gloo control plane, an
    RCCL canary in a child process per rank, agreement over gloo.  This container has no GPU, so the real canary fails on its
    own ("natural"); a crashing and a hanging canary are injected too (the hang is killed at its deadline).  Every rank must
    end on the gloo exchange with the reason on record, and the job must go on."""
    port = 35500 + (os.getpid() + len(fault) * 17 + len(prefer)) % 2000
    script = tmp_path / "worker_init.py"
    script.write_text(WORKER_INIT.format(root=ROOT, port=port))
    env = dict(os.environ, ECGPU_NCCL_PROBE_TIMEOUT="40" if not fault else "8")
    if fault:
        env["ECGPU_NCCL_PROBE_FAIL"] = fault
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", prefer], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "exchange " + kind in o, o
        if fault == "hang":
            assert "killed" in o, o


def test_canary_hosts_its_own_rendezvous_under_torchrun(monkeypatch):
    """Under `python -m torch.distributed.run` (the driver's launch) every worker inherits TORCHELASTIC_USE_AGENT_STORE=True, which
    makes torch connect to the AGENT's store as a client.  The RCCL canary has a rendezvous of its own on MASTER_PORT + 1: with that
    variable left in its environment nobody hosts the store there and every canary times out — a healthy 8-GPU node would be sent to
    the gloo fallback (the reason on record in the round-4 / round-5 dry runs).  The child's environment must carry no TORCHELASTIC_*
    variable, its own port and the device it probes."""
    ecgpu = importlib.import_module("elliptic-curves_amd")
    sharded = sys.modules[ecgpu.__name__ + ".sharded"]
    seen = {}

    class FakeProc:
        returncode = 0

        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, dict(env)

        def communicate(self, timeout=None):
            return "NCCL_PROBE_OK\n", ""

    monkeypatch.setattr(sharded.subprocess, "Popen", FakeProc)
    for k, v in {"TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x", "TORCHELASTIC_RESTART_COUNT": "0",
                 "TORCHELASTIC_MAX_RESTARTS": "0", "MASTER_PORT": "29541", "MASTER_ADDR": "127.0.0.1", "RANK": "1", "WORLD_SIZE": "2"}.items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("ECGPU_NCCL_PROBE_FAIL", raising=False)
    ok, reason = sharded.run_nccl_probe(3, timeout=50)
    assert ok and "healthy" in reason
    env = seen["env"]
    assert not [k for k in env if k.startswith("TORCHELASTIC_")], sorted(k for k in env if k.startswith("TORCHELASTIC_"))
    assert env["MASTER_PORT"] == "29542" and env["ECGPU_PROBE_DEVICE"] == "3" and env["RANK"] == "1" and env["WORLD_SIZE"] == "2"
    assert seen["cmd"][-1] == "--nccl-probe"
