"""Device-pointer entry points on torch tensors against the host-pointer ones (run by test_gpu_parity.py in a
subprocess; torch is imported first so that libecgpu.so binds to the HIP runtime torch ships)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))                # the repository root: the package directory lives there
import pyec                                            # noqa: E402
from gpu_common import ecdsa_cases, ecdsa_pack, ecgpu_module, rand_scalars   # noqa: E402

ecgpu = ecgpu_module()
eng = ecgpu.Engine(0)
c = pyec.K256
L, n = c.L, 4096
dev = "cuda:0"
k = rand_scalars(c.cid, n, 0xEC0000D7)
k2 = rand_scalars(c.cid, n, 0xEC0000D8)
want_xy, want_inf = eng.mul_by_generator(c.cid, k)
d_k = torch.from_numpy(np.ascontiguousarray(k)).to(dev)
d_k2 = torch.from_numpy(np.ascontiguousarray(k2)).to(dev)
d_out = torch.empty((n, 2 * L), dtype=torch.uint8, device=dev)
d_inf = torch.empty((n + 16,), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()          # the engine's stream is not ordered after torch's (include/ecgpu.h)
eng.mul_by_generator_dev(c.cid, d_k, n, d_out, d_inf)
assert bytes(d_out.cpu().numpy().reshape(-1)) == bytes(want_xy) and bytes(d_inf[:n].cpu().numpy()) == bytes(want_inf)
d_x = torch.empty((n, L), dtype=torch.uint8, device=dev)
d_tag = torch.empty((n + 16,), dtype=torch.uint8, device=dev)
eng.mul_by_generator_compressed_dev(c.cid, d_k, n, d_x, d_tag)
wx, wt = eng.mul_by_generator_compressed(c.cid, k)
assert bytes(d_x.cpu().numpy().reshape(-1)) == bytes(wx) and bytes(d_tag[:n].cpu().numpy()) == bytes(wt)
d_pts = d_out.clone()
torch.cuda.synchronize()
eng.mul_dev(c.cid, d_k2, d_pts, None, n, d_out, d_inf)
w2, wi2 = eng.mul(c.cid, k2, want_xy)
assert bytes(d_out.cpu().numpy().reshape(-1)) == bytes(w2)
d_r = torch.empty((1, 2 * L), dtype=torch.uint8, device=dev)
d_ri = torch.empty((16,), dtype=torch.uint8, device=dev)
eng.lincomb_dev(c.cid, d_k2, d_pts, None, n, d_r, d_ri)
w3, wf3 = eng.lincomb(c.cid, k2, want_xy)
assert bytes(d_r.cpu().numpy().reshape(-1)) == bytes(w3) and int(d_ri[0].item()) == wf3
d_sx = torch.empty((n, L), dtype=torch.uint8, device=dev)
d_ok = torch.empty((n + 16,), dtype=torch.uint8, device=dev)
eng.ecdh_dev(c.cid, d_k2, d_pts, n, d_sx, d_ok)
w4, ok4 = eng.ecdh(c.cid, k2, want_xy)
assert bytes(d_sx.cpu().numpy().reshape(-1)) == bytes(w4) and bytes(d_ok[:n].cpu().numpy()) == bytes(ok4)
z, r, s_, q, exp = ecdsa_pack(ecdsa_cases(c, 0xE2))
m = len(exp)
t = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).to(dev)
d_z, d_rr, d_s, d_q = t(z), t(r), t(s_), t(q)
torch.cuda.synchronize()
eng.ecdsa_verify_dev(c.cid, d_z, d_rr, d_s, d_q, m, False, d_ok)
assert bytes(d_ok[:m].cpu().numpy()) == bytes(exp)
# the same with the engine on torch's current stream: no synchronisation needed then
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    eng.set_stream(s1.cuda_stream)
    d_k3 = (d_k.clone() ^ 0)                 # produced on s1
    eng.mul_by_generator_dev(c.cid, d_k3, n, d_out, d_inf)
    eng.set_stream(None)
assert bytes(d_out.cpu().numpy().reshape(-1)) == bytes(want_xy)
print("device-pointer entry points: ok")
