"""Bisecting probes for ecgpu_ecdsa_recover_batch on one curve: choices of (z, s) that make the scalars a = -(z/r), b = s/r
trivial, so that a wrong key points at the in-kernel decompression, the inversion, or the products."""
import importlib, os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyec
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
curve = sys.argv[1] if len(sys.argv) > 1 else "p521"
c = pyec.CURVES[curve]
L = c.L
rng = random.Random(7)
G = pyec.G(c)
rs = []
while len(rs) < 48:
    x = rng.randrange(1, c.n)
    if pyec.lift_x(c, x, 0) is not None:
        rs.append(x)
enc = lambda v: b"".join(x.to_bytes(L, "big") for x in v)
ids = bytes([i & 1 for i in range(len(rs))])
Rs = [pyec.lift_x(c, x, i & 1) for i, x in enumerate(rs)]
def probe(name, zs, ss, want):
    out, ok = e.ecdsa_recover(c.cid, enc(zs), enc(rs), enc(ss), ids)
    bad = [i for i in range(len(rs)) if not ok[i] or bytes(out[2 * L * i: 2 * L * (i + 1)]) != pyec.enc_point(c, want[i])[0]]
    notok = [i for i in bad if not ok[i]]
    print("%-34s bad %2d / %d  (ok = 0: %d)  %s" % (name, len(bad), len(rs), len(notok), bad[:12]), flush=True)
    return bad
inv = [pow(x, -1, c.n) for x in rs]
probe("z=0 s=r: key = R", [0] * len(rs), rs, Rs)
probe("z=0 s=1: key = (1/r) R", [0] * len(rs), [1] * len(rs), [pyec.mul(c, inv[i], Rs[i]) for i in range(len(rs))])
probe("z=r s=r: key = R - G", rs, rs, [pyec.add(c, Rs[i], pyec.neg(c, G)) for i in range(len(rs))])
probe("z=1 s=r: key = R - (1/r) G", [1] * len(rs), rs, [pyec.add(c, Rs[i], pyec.neg(c, pyec.mul(c, inv[i], G))) for i in range(len(rs))])
z2 = [rng.randrange(c.n) for _ in rs]
bad = probe("z random s=r: key = R - (z/r) G", z2, rs, [pyec.add(c, Rs[i], pyec.neg(c, pyec.mul(c, inv[i] * z2[i] % c.n, G))) for i in range(len(rs))])
s2 = [rng.randrange(1, c.n) for _ in rs]
probe("z=0 s random: key = (s/r) R", [0] * len(rs), s2, [pyec.mul(c, inv[i] * s2[i] % c.n, Rs[i]) for i in range(len(rs))])
# what do the failing elements have in common?
for i in range(12):
    print("   i=%d %s r=%s.. z=%s.. bits(r)=%d bits(z)=%d bits(1/r)=%d" % (i, "BAD" if i in bad else "ok ", hex(rs[i])[:12], hex(z2[i])[:12], rs[i].bit_length(), z2[i].bit_length(), inv[i].bit_length()))
# the verification kernel on the same population, for comparison (u1 = z/s, u2 = r/s)
d = [rng.randrange(1, c.n) for _ in rs]
k = [rng.randrange(1, c.n) for _ in rs]
sig = [pyec.ecdsa_sign(c, d[i], z2[i], k[i]) for i in range(len(rs))]
Q = b"".join(pyec.enc_point(c, pyec.mul(c, d[i], G))[0] for i in range(len(rs)))
okv = e.ecdsa_verify(c.cid, enc(z2), enc([t[0] for t in sig]), enc([t[1] for t in sig]), Q)
print("ecdsa_verify on %d valid signatures: %d accepted" % (len(rs), int(okv.sum())))
