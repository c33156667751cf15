"""Checks every limb constant in csrc/ecgpu_field.h against the big-integer curve parameters."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "elliptic-curves_amd", "csrc", "ecgpu_params.h")


def arrays(src, struct):
    start = src.index("struct %s {" % struct)
    body = src[start: src.index("\n};", start)]
    out = {}
    for m in re.finditer(r"uint32_t (\w+)\[\d+\] = \{([^}]*)\}", body):
        limbs = [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")]
        out[m.group(1)] = sum(l << (32 * i) for i, l in enumerate(limbs))
    return out


def check():
    src = open(HEADER).read()
    bad = []
    for name, c in (("K256Params", pyec.K256), ("P256Params", pyec.P256), ("P384Params", pyec.P384)):
        a = arrays(src, name)
        R = 1 << (8 * c.L)
        exp = {"P": c.p, "ORDER": c.n, "GX": c.gx, "GY": c.gy, "ORDER_R2": R * R % c.n}
        m = re.search(r"ORDER_NINV32 = 0x([0-9A-Fa-f]+)u", src[src.index("struct %s {" % name):])
        if not m or (int(m.group(1), 16) * c.n + 1) % (1 << 32) != 0:
            bad.append((name, "ORDER_NINV32"))
        if name == "K256Params":
            exp["BETA"] = pyec.K256_BETA if hasattr(pyec, "K256_BETA") else a.get("BETA")
        if name != "K256Params":
            exp.update({"R2": R * R % c.p, "ONE": R % c.p, "B": c.b})
        for k, v in exp.items():
            if a.get(k) != v:
                bad.append((name, k, hex(a.get(k, -1)), hex(v)))
    return bad


if __name__ == "__main__":
    bad = check()
    for b in bad:
        print("MISMATCH", b)
    print("ok" if not bad else "FAILED")
