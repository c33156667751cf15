"""Checks every limb constant in csrc/ecgpu_field.h against the big-integer curve parameters."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "elliptic-curves_amd", "csrc", "ecgpu_params.h")


def arrays(src, struct):
    start = re.search(r"struct %s\b[^{]*\{" % struct, src).start()
    body = src[start: src.index("\n};", start)]
    out = {}
    for m in re.finditer(r"uint32_t (\w+)\[\d+\] = \{([^}]*)\}", body):
        limbs = [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")]
        out[m.group(1)] = sum(l << (32 * i) for i, l in enumerate(limbs))
    return out


SETS = (("K256Params", "k256"), ("P256Params", "p256"), ("P384Params", "p384"), ("Sm2Params", "sm2"), ("P224Params", "p224"),
        ("P192Params", "p192"), ("P521Params", "p521"), ("Bp256Params", "bp256"), ("Bp384Params", "bp384"),
        ("Bp256t1Params", "bp256t1"), ("Bp384t1Params", "bp384t1"), ("Bign256Params", "bign256"))


def check():
    """Every constant array a parameter set declares (P, ORDER, B, GX, GY, ORDER_R2, ORDER_NINV32, BETA, ...) against the
    big-integer curve parameters of tests/pyec.py — all twelve sets; a struct that inherits (the t1 twists) is checked on
    what it overrides."""
    src = open(HEADER).read()
    bad = []
    for name, key in SETS:
        c = pyec.CURVES[key]
        a = arrays(src, name)
        body = src[src.index("struct %s " % name):]
        body = body[: body.index("\n};")]
        nwords = int(re.search(r"uint32_t (?:P|B)\[(\d+)\]", body).group(1))
        R = 1 << (32 * nwords)
        exp = {"P": c.p, "ORDER": c.n, "GX": c.gx, "GY": c.gy, "ORDER_R2": R * R % c.n, "B": c.b % c.p,
               "R2": R * R % c.p, "ONE": R % c.p}
        if name == "K256Params":
            exp["BETA"] = pyec.K256_BETA
            exp.pop("B")                                  # k256 carries b = 7 as a small constant, not an array
        m = re.search(r"ORDER_NINV32 = 0x([0-9A-Fa-f]+)u", body)
        if m and (int(m.group(1), 16) * c.n + 1) % (1 << 32) != 0:
            bad.append((name, "ORDER_NINV32"))
        if not m and ":" not in body.split("{")[0]:        # only a derived struct may leave it to its base
            bad.append((name, "ORDER_NINV32 missing"))
        seen = 0
        for k, v in exp.items():
            if k in a:
                seen += 1
                if a[k] != v:
                    bad.append((name, k, hex(a[k]), hex(v)))
        if seen < 3:
            bad.append((name, "only %d constants found" % seen))
    return bad


if __name__ == "__main__":
    bad = check()
    for b in bad:
        print("MISMATCH", b)
    print("ok" if not bad else "FAILED")
