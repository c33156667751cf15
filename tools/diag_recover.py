"""One-off diagnosis of ecgpu_ecdsa_recover_batch on a curve: which element fails, and which stage (decompression, the
scalars, the a G + b R kernels) disagrees with the big-int model.  python tools/diag_recover.py [curve] [seed]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyec, gpu_common
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
curve = sys.argv[1] if len(sys.argv) > 1 else "p521"
c = pyec.CURVES[curve]
L = c.L
seed = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x4EC2 + c.cid
cases = gpu_common.recover_cases(c, seed, nvalid=12)
z, r, s, recid, exy, eok = gpu_common.recover_pack(cases, L)
m = len(cases)
for rep in range(2):
    out, ok = e.ecdsa_recover(c.cid, z, r, s, recid)
    bad = [i for i in range(m) if ok[i] != eok[i] or bytes(out[2 * L * i: 2 * L * (i + 1)]) != exy[2 * L * i: 2 * L * (i + 1)]]
    print(curve, "batch of", m, "run", rep, "bad", bad, flush=True)
# reversed order: does the failure follow the data or the position?
perm = list(range(m))[::-1]
pz, pr, ps = (b"".join(b[L * i: L * i + L] for i in perm) for b in (z, r, s))
out2, ok2 = e.ecdsa_recover(c.cid, pz, pr, ps, np.array([recid[i] for i in perm], np.uint8))
bad2 = [perm[j] for j in range(m) if ok2[j] != eok[perm[j]] or bytes(out2[2 * L * j: 2 * L * (j + 1)]) != exy[2 * L * perm[j]: 2 * L * (perm[j] + 1)]]
print("reversed batch: bad (original indices)", sorted(bad2), flush=True)
for i in bad[:4]:
    zi, ri, si = (int.from_bytes(b[L * i: L * i + L], "big") for b in (z, r, s))
    rid = int(recid[i])
    o1, k1 = e.ecdsa_recover(c.cid, z[L * i: L * i + L], r[L * i: L * i + L], s[L * i: L * i + L], bytes([rid]))
    print(" idx", i, "recid", rid, "expect ok", int(eok[i]), "batch ok", int(ok[i]), "alone ok", int(k1[0]), "alone key right", bytes(o1) == exy[2 * L * i: 2 * L * (i + 1)])
    x = ri + (c.n if rid & 2 else 0)
    R = pyec.lift_x(c, x, rid & 1) if x < 1 << (8 * L) else None
    if x < 1 << (8 * L):
        dxy, dok = e.decompress(c.cid, x.to_bytes(L, "big"), bytes([rid & 1]))
        print("   decompress ok", int(dok[0]), "model", R is not None, "same", R is None or bytes(dxy) == pyec.enc_point(c, R)[0])
    if R is not None:
        rinv = pow(ri, -1, c.n)
        a, b = (-rinv * (zi % c.n)) % c.n, rinv * si % c.n
        kxy, kinf = e.mul_by_generator_and_mul_add(c.cid, a.to_bytes(L, "big"), b.to_bytes(L, "big"), pyec.enc_point(c, R)[0])
        print("   a G + b R with host scalars: right", bytes(kxy) == exy[2 * L * i: 2 * L * (i + 1)], "inf", int(kinf[0]))
        print("   verify with the expected key", int(e.ecdsa_verify(c.cid, z[L * i: L * i + L], r[L * i: L * i + L], s[L * i: L * i + L], exy[2 * L * i: 2 * L * (i + 1)])[0]))
        # the same signature under the other ids
        for rid2 in range(4):
            o3, k3 = e.ecdsa_recover(c.cid, z[L * i: L * i + L], r[L * i: L * i + L], s[L * i: L * i + L], bytes([rid2]))
            Q = pyec.ecdsa_recover(c, zi, ri, si, rid2)
            print("   id", rid2, "device ok", int(k3[0]), "model ok", Q is not None, "key same", (Q is None and not k3[0]) or (Q is not None and bytes(o3) == pyec.enc_point(c, Q)[0]))
# a bigger random population: how often does it fail?
import random
rng = random.Random(99)
G = pyec.G(c)
zs = rs = ss = b""; ids = []; exp = []
for t in range(96):
    d, k, zz = rng.randrange(1, c.n), rng.randrange(1, c.n), rng.randrange(c.n)
    Rk = pyec.mul(c, k, G); rr, s_ = pyec.ecdsa_sign(c, d, zz, k)
    zs += zz.to_bytes(L, "big"); rs += rr.to_bytes(L, "big"); ss += s_.to_bytes(L, "big"); ids.append(Rk[1] & 1); exp.append(pyec.enc_point(c, pyec.mul(c, d, G))[0])
o4, k4 = e.ecdsa_recover(c.cid, zs, rs, ss, bytes(ids))
bad4 = [t for t in range(96) if not k4[t] or bytes(o4[2 * L * t: 2 * L * (t + 1)]) != exp[t]]
print("96 random valid signatures: bad", bad4, flush=True)
for t in bad4[:3]:
    print("   r top bytes", rs[L * t: L * t + 4].hex(), "s top", ss[L * t: L * t + 4].hex(), "z top", zs[L * t: L * t + 4].hex(), "ok", int(k4[t]))
