import importlib, os, random, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pyec
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
for curve in ("p521",):
    c = pyec.CURVES[curve]
    rng = random.Random(5)
    vals = [rng.randrange(c.p) for _ in range(512)]
    oth = [rng.randrange(c.p) for _ in range(512)]
    fe = lambda v: np.frombuffer(b"".join(x.to_bytes(c.L, "big") for x in v), np.uint8)
    A, B = fe(vals), fe(oth)
    p = c.p
    want = {0: [(a + b) % p for a, b in zip(vals, oth)], 1: [(a - b) % p for a, b in zip(vals, oth)], 2: [a * b % p for a, b in zip(vals, oth)],
            3: [a * a % p for a in vals], 5: [(-a) % p for a in vals], 7: [2 * a % p for a in vals], 8: [(2 * a + b) % p for a, b in zip(vals, oth)],
            9: [(-b * b) % p for a, b in zip(vals, oth)], 20: vals, 21: vals}
    for op, w in want.items():
        out = bytes(e.selftest_field(c.cid, op, A, B))
        got = [int.from_bytes(out[i * c.L:(i + 1) * c.L], "big") for i in range(512)]
        bad = [(i, g, x) for i, (g, x) in enumerate(zip(got, w)) if g != x]
        print(curve, "op", op, "bad", len(bad), [hex((g - x) % p)[:24] + ".." + str(((g - x) % p).bit_length()) for _, g, x in bad[:4]],
              [("a+b>=p" if vals[i] + oth[i] >= p else "a+b<p") for i, _, _ in bad[:4]], flush=True)
