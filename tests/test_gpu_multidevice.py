"""First-contact tests for boxes with MORE THAN ONE GPU (-m gpu; every test here skips cleanly below two devices).

The development box and the round-end test box have one MI355X, so until an 8-GPU node runs this file the paths below have
only been exercised with several contexts on ONE device (tests/test_gpu_parity.py::test_group_multi_device_entry_points) and
over gloo on CPU (tests/test_sharded_gloo.py).  What only distinct devices can show:
  * `torch.distributed` backend "nccl" (RCCL over xGMI) moving the per-window partial sums of the sharded MSM
    (sharded.RecordExchange.all_gather_into_tensor) — tests/mgpu_worker.py, one process per GPU, the launch bench.py gets;
  * `ecgpu_group_init({0, 1, ...})` with RCCL's ncclCommInitAll / ncclAllGather from ONE process, and the peer-copy
    exchange (hipMemcpyPeerAsync into GPU 0) between distinct devices;
  * unequal and empty shards on real devices.
Reference behaviour to match: `lincomb` (k256/src/arithmetic/mul.rs:84-109, primeorder/src/projective.rs:480-511) — the
bytes of the single-GPU result and of the oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib
import pyec
from gpu_common import ecgpu_module, rand_scalars

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def need_gpus(k):
    n = device_count()
    if n < k:
        pytest.skip("needs %d GPUs, this box has %d" % (k, n))
    return n


@pytest.fixture(scope="module")
def eng():
    e = ecgpu_module().Engine(0)
    yield e
    e.close()


def worlds():
    n = device_count()
    return [w for w in (2, 4, 8) if w <= n] or [2]


@pytest.mark.parametrize("world", worlds())
def test_nccl_ranks_sharded_msm_equals_single_gpu(world):
    """One process per GPU under torch.distributed.run, RCCL all-gather of the parts: every rank's result == the
    single-GPU MSM == the exact dot product (tests/mgpu_worker.py has the cases)."""
    need_gpus(world)
    r = run_workers(world, MGPU_REQUIRE_RCCL="1")
    assert r.returncode == 0 and "MGPU_WORKER_OK world=%d exchange=rccl" % world in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def run_workers(world, **extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra)
    port = 29700 + (os.getpid() + 7 * world + len(extra)) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "mgpu_worker.py")]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("fault", ["natural", "crash", "hang"])
def test_two_ranks_fall_back_to_gloo_when_rccl_is_not_to_be_had(fault):
    """What the driver's one scaling run must survive, reproduced on ONE GPU: two ranks share device 0, so RCCL cannot form
    its communicator (`natural`: measured in round 4, its init neither fails nor returns — the canary process of
    sharded.init_exchange is killed at its deadline, which is exactly the case a hung main process could not report), or the
    canary is made to crash / hang (fault injection: ECGPU_NCCL_PROBE_FAIL).  Either way every rank agrees on the gloo
    exchange, the sharded MSM results are right on every rank, and the reason is on record."""
    if device_count() < 1:
        pytest.skip("needs a GPU")
    extra = dict(MGPU_SHARE_GPU="1")
    if fault != "natural":
        extra.update(ECGPU_NCCL_PROBE_FAIL=fault, ECGPU_NCCL_PROBE_TIMEOUT="25")
    else:
        extra.update(ECGPU_NCCL_PROBE_TIMEOUT="45")      # (RCCL with two ranks on one device does not fail, it HANGS: the canary is killed)
    r = run_workers(2, **extra)
    assert r.returncode == 0 and "MGPU_WORKER_OK world=2 exchange=gloo-fallback" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    if fault == "hang":
        assert "killed" in r.stdout


@pytest.mark.parametrize("mode", ["rccl", "peer"])
def test_group_on_distinct_devices(eng, mode, monkeypatch):
    """ecgpu_group_* over every GPU of the box (and over the first two), both exchanges: lincomb with an even split, an uneven
    one, fewer terms than GPUs (empty shards) and no terms; the batch calls' slices.  Results == member-less single-GPU calls
    == the oracle."""
    ndev = need_gpus(2)
    ecgpu = ecgpu_module()
    for devices in ([0, 1], list(range(ndev)), list(range(ndev))[::-1]):
        try:
            grp = ecgpu.Group(devices, exchange=mode)
        except ecgpu.EcgpuError:
            if mode == "rccl":
                pytest.skip("librccl could not be loaded / initialised in this process")
            raise
        try:
            assert grp.size == len(devices) and grp.exchange == mode
            for curve in ("k256", "p256", "p384"):
                c = pyec.CURVES[curve]
                for n in ((1 << 17) + 5 * len(devices) + 1, 1 << 15, len(devices) - 1, 1, 0):
                    k = rand_scalars(c.cid, n, 0xEC0009F7 + c.cid + n)
                    s = rand_scalars(c.cid, n, 0xEC000AF7 + c.cid + n)
                    pts, _ = eng.mul_by_generator(c.cid, s)
                    pts = pts.copy()
                    inf = np.zeros(n, np.uint8)
                    inf[::7] = 1
                    pts.reshape(n, 2 * c.L)[::7] = 0
                    want, wf = eng.lincomb(c.cid, k, pts, inf)
                    got, gf = grp.lincomb(c.cid, k, pts, inf)
                    assert bytes(got) == bytes(want) and gf == wf, (curve, n, devices, mode)
                    if n <= (1 << 15):
                        o, of = oracle_lib.msm(c.cid, k, pts, inf, vartime=True)
                        assert bytes(got) == bytes(o) and gf == of
                n = 4099
                k = rand_scalars(c.cid, n, 0xEC000BF7 + c.cid)
                a, ai = grp.mul_by_generator(c.cid, k)
                b, bi = eng.mul_by_generator(c.cid, k)
                assert bytes(a) == bytes(b) and bytes(ai) == bytes(bi)
                k2 = rand_scalars(c.cid, n, 0xEC000CF7 + c.cid)
                a2, ai2 = grp.mul(c.cid, k2, b)
                b2, bi2 = eng.mul(c.cid, k2, b)
                assert bytes(a2) == bytes(b2) and bytes(ai2) == bytes(bi2)
            # a window override on the group reaches every member (one plan for all: the parts records must agree)
            grp.set_msm_window(12)
            c = pyec.CURVES["k256"]
            n = 70001
            k = rand_scalars(c.cid, n, 0xEC000DF7)
            pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xEC000EF7))
            want, wf = eng.lincomb(c.cid, k, pts)
            got, gf = grp.lincomb(c.cid, k, pts)
            assert bytes(got) == bytes(want) and gf == wf
            grp.set_msm_window(0)
        finally:
            grp.close()


def test_group_rejects_diverging_member_plans(eng, monkeypatch):
    """A caller who changes the Pippenger window of ONE member through ecgpu_group_ctx would make that member write a
    different-sized parts record: the group must refuse (ECGPU_ERR_ARG), not overflow.  Runs on one GPU too (two contexts on
    device 0)."""
    import ctypes
    ecgpu = ecgpu_module()
    ndev = device_count()
    grp = ecgpu.Group([0, 1] if ndev >= 2 else [0, 0], exchange="peer")
    try:
        lib = ecgpu.load_library()
        lib.ecgpu_group_ctx.restype = ctypes.c_void_p
        ctx1 = ctypes.c_void_p(lib.ecgpu_group_ctx(grp._g, 1))
        assert lib.ecgpu_set_msm_window(ctx1, 11) == 0
        c = pyec.CURVES["k256"]
        n = 1 << 17
        k = rand_scalars(c.cid, n, 0xEC000FF7)
        pts, _ = eng.mul_by_generator(c.cid, rand_scalars(c.cid, n, 0xEC0010F7))
        with pytest.raises(ecgpu.EcgpuError) as ei:
            grp.lincomb(c.cid, k, pts)
        assert ei.value.code == ecgpu.ERR_ARG
        assert lib.ecgpu_set_msm_window(ctx1, 0) == 0
        want, wf = eng.lincomb(c.cid, k, pts)
        got, gf = grp.lincomb(c.cid, k, pts)
        assert bytes(got) == bytes(want) and gf == wf
    finally:
        grp.close()


def test_comb_table_is_shared_per_device_and_outlives_its_builder():
    """One comb table per (device, curve, width) for the whole process: a second context on the same GPU finds the table
    the first one built (no second 21.5 GB allocation, no second build) and keeps it alive after the first is closed."""
    import time
    import torch
    ecgpu = ecgpu_module()
    c = pyec.CURVES["k256"]
    k = rand_scalars(c.cid, 257, 0xEC0011F7)
    want, winf = oracle_lib.batch_mul_base(c.cid, k)
    a = ecgpu.Engine(0)
    out, inf = a.mul_by_generator(c.cid, k)                       # builds (or finds) the table
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    b = ecgpu.Engine(0)
    t0 = time.perf_counter()
    out, inf = b.mul_by_generator(c.cid, k)
    dt = time.perf_counter() - t0
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    used = free_before - torch.cuda.mem_get_info(0)[0]
    assert used < (2 << 30), "the second context allocated %d bytes: its own table?" % used
    a.close()
    out, inf = b.mul_by_generator(c.cid, k)                       # the table is still there
    assert bytes(out) == bytes(want)
    # a different width on one context is a different table; the other context keeps its own view
    b.set_base_window(c.cid, 13)
    out, inf = b.mul_by_generator(c.cid, k)
    assert bytes(out) == bytes(want)
    b.close()
    assert dt < 5.0


def test_comb_table_falls_back_to_a_narrower_window_when_it_does_not_fit(monkeypatch):
    """The test hook ecgpu_testhook_table_max_mb (an exported symbol that is not in include/ecgpu.h; no environment variable can
    steer the production path) makes the table allocation refuse anything larger (fault injection: a real out-of-memory needs
    a full GPU): the width drops two bits at a time, results stay identical, a second context of the device goes straight to
    the width that worked; when not even 16 bits fit the call returns ECGPU_ERR_OOM (not ECGPU_ERR_HIP) and the context
    stays usable."""
    import ctypes
    ecgpu = ecgpu_module()
    hook = ecgpu.load_library().ecgpu_testhook_table_max_mb
    hook.restype, hook.argtypes = None, [ctypes.c_size_t]
    c = pyec.CURVES["p192"]
    k = rand_scalars(c.cid, 300, 0xEC0012F7)
    want, winf = oracle_lib.batch_mul_base(c.cid, k)
    e = ecgpu.Engine(0)
    try:
        e.set_base_window(c.cid, 22)          # 2^21 x 9 windows x 48 B = 906 MB; 20: 252 MB; 18: 69 MB; 16: 19 MB
        hook(64)
        out, inf = e.mul_by_generator(c.cid, k)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
        e2 = ecgpu.Engine(0)                  # the registry remembers the refused widths: no second attempt at 22 / 20 / 18
        try:
            e2.set_base_window(c.cid, 22)
            out, inf = e2.mul_by_generator(c.cid, k)
            assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
        finally:
            e2.close()
        hook(8)
        e.set_base_window(c.cid, 21)          # 21 -> 19 -> 17: 38 MB; 15 is below the floor
        with pytest.raises(ecgpu.EcgpuError) as ei:
            e.mul_by_generator(c.cid, k)
        assert ei.value.code == ecgpu.ERR_OOM
        hook(0)
        e.set_base_window(c.cid, 12)
        out, inf = e.mul_by_generator(c.cid, k)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    finally:
        hook(0)
        e.close()


def test_a_context_on_a_narrower_table_gets_the_wide_one_when_the_refusal_expires():
    """include/ecgpu.h: "a table that does not fit is replaced by one two bits narrower" — and that refusal is not for ever.  A
    context that fell back keeps asking: while the refusal stands (here: the test hook's cap; 64 calls per attempt) it stays on
    the narrow table — a second refusal costs one failed allocation and nothing else —, and once the memory is there the wide
    table is built BEFORE the narrow one is let go and the context moves over.  Results never change."""
    import ctypes
    ecgpu = ecgpu_module()
    hook = ecgpu.load_library().ecgpu_testhook_table_max_mb
    hook.restype, hook.argtypes = None, [ctypes.c_size_t]
    c = pyec.CURVES["p192"]
    k = rand_scalars(c.cid, 300, 0xEC0013F7)
    want, winf = oracle_lib.batch_mul_base(c.cid, k)
    e = ecgpu.Engine(0)
    try:
        e.set_base_window(c.cid, 22)          # 906 MB; the cap lets 16 bits (19 MB) through
        hook(64)
        out, inf = e.mul_by_generator(c.cid, k)
        assert bytes(out) == bytes(want) and e.base_table_info(c.cid)["window_bits"] == 16
        for _ in range(150):                  # two expiries of the refusal with the cap still in place: retried, refused again
            out, inf = e.mul_by_generator(c.cid, k)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf) and e.base_table_info(c.cid)["window_bits"] == 16
        hook(0)                               # the memory is there now
        for _ in range(70):
            out, inf = e.mul_by_generator(c.cid, k)
        assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
        assert e.base_table_info(c.cid)["window_bits"] == 22, e.base_table_info(c.cid)
    finally:
        hook(0)
        e.close()


def test_one_rank_rccl_exchange_is_ordered_between_the_halves_on_the_device():
    """tests/gpu_rccl_one_rank_check.py: bench.py's N > 1 MSM step with a REAL RCCL all-gather (a world of one rank: the one-GPU box
    can show this), queued back to back without a host wait — every step must combine ITS record — and the same on two lanes."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_rccl_one_rank_check.py")],
                       capture_output=True, text=True, timeout=600)
    if "SKIP:" in r.stdout:
        pytest.skip(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

