"""One rank, REAL RCCL (tests/test_gpu_multidevice.py runs this in a process of its own; a one-GPU box suffices).

What bench.py's N > 1 MSM step relies on and a one-GPU box CAN show: the all-gather of the parts record runs on torch's collective
stream, the engine queues on torch's current stream (ecgpu_set_stream), and NOTHING waits on the host between the local half, the
collective and the combining half (`RecordExchange.gather(consumer_on_current_stream=True)`, ecgpu_set_async).  If the collective were
not ordered between the two halves on the device, a step would combine the previous step's record: every step here has different
scalars, the steps are queued back to back without a host wait, and every result is compared with the one-call MSM of its inputs.
Then the same through the throughput form: local halves on two rotating lanes, ecgpu_msm_parts_join_dev before each collective.
A world of one rank is an all-gather of one record — the stream hand-over is the same as with eight."""
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
    torch.cuda.set_device(0)
    dev = "cuda:0"
    port = 36000 + os.getpid() % 2000
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device(dev))
    except Exception as e:                                            # no RCCL in this process: nothing to show
        print("SKIP: nccl process group of one rank: %s" % str(e).splitlines()[0][:200])
        return 0
    eng = ecgpu.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    oracle_lib.build()
    for cid, n in ((ecgpu.K256, (1 << 18) + 5), (ecgpu.K256, 1 << 21), (ecgpu.P256, (1 << 17) + 1)):
        L = ecgpu.FIELD_BYTES[cid]
        order = ecgpu.GROUP_ORDERS[cid]
        steps = 6
        rng = np.random.default_rng(0xEC0081F7 + cid + n)
        s = oracle_lib.scalar_reduce(cid, rng.integers(0, 256, n * L, dtype=np.uint8))
        d_s = torch.from_numpy(s.copy()).to(dev).reshape(n, L)
        d_pts = torch.empty((n, 2 * L), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.mul_by_generator_dev(cid, d_s, n, d_pts, None)
        ks = [oracle_lib.scalar_reduce(cid, rng.integers(0, 256, n * L, dtype=np.uint8)) for _ in range(steps)]
        d_ks = [torch.from_numpy(k.copy()).to(dev).reshape(n, L) for k in ks]
        want = []
        for k in ks:                                                  # (sum k_i s_i mod n) G: exact, whatever the term count
            dot = sum(int.from_bytes(k[i * L:(i + 1) * L].tobytes(), "big") * int.from_bytes(s[i * L:(i + 1) * L].tobytes(), "big")
                      for i in range(0, n)) % order if n <= (1 << 18) + 5 else None
            o = torch.zeros((1, 2 * L), dtype=torch.uint8, device=dev)
            f = torch.zeros((16,), dtype=torch.uint8, device=dev)
            eng.lincomb_dev(cid, torch.from_numpy(k.copy()).to(dev).reshape(n, L), d_pts, None, n, o, f)
            torch.cuda.synchronize()
            got = bytes(o.cpu().numpy().reshape(-1)) + bytes([int(f[0].item())])
            if dot is not None:
                w, wf = oracle_lib.batch_mul_base(cid, np.frombuffer(dot.to_bytes(L, "big"), np.uint8))
                assert got == bytes(w) + bytes([int(wf[0])]), "one-call MSM != (sum k s) G"
            want.append(got)
        nbytes = eng.msm_parts_bytes(cid, n)
        for lanes in (1, 2):
            exs = [ecgpu.RecordExchange(torch, dist, nbytes, dev) for _ in range(lanes)]
            outs = [(torch.zeros((1, 2 * L), dtype=torch.uint8, device=dev), torch.zeros((16,), dtype=torch.uint8, device=dev)) for _ in range(steps)]
            eng.set_async(True)
            if lanes > 1:
                eng.set_msm_lanes(lanes)
            pend = []

            def combine():
                i = pend.pop(0)
                ex = exs[i % lanes]
                eng.msm_parts_join_dev(ex.mine)
                eng.msm_finish_dev(cid, ex.gather(consumer_on_current_stream=True), 1, n, *outs[i])

            for i in range(steps):                                    # queued back to back: no host wait anywhere in this loop
                eng.msm_parts_dev(cid, d_ks[i], d_pts, None, n, n, exs[i % lanes].mine)
                pend.append(i)
                if lanes == 1 or len(pend) > 1:
                    combine()
            while pend:
                combine()
            eng.synchronize()
            if lanes > 1:
                eng.set_msm_lanes(1)
            eng.set_async(False)
            torch.cuda.synchronize()
            for i in range(steps):
                got = bytes(outs[i][0].cpu().numpy().reshape(-1)) + bytes([int(outs[i][1][0].item())])
                assert got == want[i], "curve %d, n %d, lanes %d: step %d combined another step's record" % (cid, n, lanes, i)
    eng.close()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
