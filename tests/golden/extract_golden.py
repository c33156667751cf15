#!/usr/bin/env python3
"""Extract the reference's known-answer vectors for the scalar-mul hot path into JSON fixtures.

Run in the build container (needs /root/reference); the GPU box only ever reads the JSON.

    python tests/golden/extract_golden.py [/root/reference]

Sources:
  {k256,p256,p384,p224,p192,p521,bignp256}/src/test_vectors/group.rs   ADD_TEST_VECTORS (k*G for k = 1..20, affine x,y)
                                               MUL_TEST_VECTORS ((k, x, y) with k*G = (x, y))
  {k256,p256,p384,p224,p192,p521}/src/test_vectors/ecdsa.rs   FIPS 186-4 style (d, Qx, Qy, k, m, r, s)
  {k256,p256}/src/test_vectors/field.rs        DBL_TEST_VECTORS (repeated doubling of 1 mod p)
  k256/src/ecdsa.rs                            RECOVERY_TEST_VECTORS (:190-211: compressed key, message, signature, recovery
                                               id) and the Ethereum end-to-end example (:233-261: signing key, RLP message
                                               hashed with Keccak-256, signature, recovery id 0)
  k256/src/schnorr.rs                          BIP340_SIGN_VECTORS (index 0-3: public key, message, valid signature),
                                               BIP340_VERIFY_VECTORS (index 4-14: public key, message, signature,
                                               expected verdict) and the variable-length-message vectors 15-18

Only the hex constants are taken (public NIST / point-at-infinity.org / FIPS 186-4 data); no
reference source code is copied.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
HEX = re.compile(r'hex!\(\s*"([0-9A-Fa-f\s]+)"\s*\)')


def const_block(text, name):
    """Text of `pub const NAME ... = &[ ... ];`"""
    m = re.search(r"pub const %s\b" % name, text)
    if not m:
        return None
    start = text.index("=", m.end())
    # the const ends at the first "];" at column 0 after start
    end = re.search(r"^\]\)?;|\}\];", text[start:], re.M)
    return text[start:start + end.end()]


def hexes(block):
    return [re.sub(r"\s+", "", h).lower() for h in HEX.findall(block)]


def group_vectors(curve):
    text = open(os.path.join(REF, curve, "src/test_vectors/group.rs")).read()
    add = hexes(const_block(text, "ADD_TEST_VECTORS"))
    mul = hexes(const_block(text, "MUL_TEST_VECTORS"))
    assert len(add) % 2 == 0 and len(mul) % 3 == 0
    return {
        "add": [{"k": i // 2 + 1, "x": add[i], "y": add[i + 1]} for i in range(0, len(add), 2)],
        "mul": [{"k": mul[i], "x": mul[i + 1], "y": mul[i + 2]} for i in range(0, len(mul), 3)],
    }


def ecdsa_vectors(curve):
    text = open(os.path.join(REF, curve, "src/test_vectors/ecdsa.rs")).read()
    out = []
    for body in re.findall(r"TestVector\s*\{(.*?)\}", text, re.S):
        fields = dict(re.findall(r'(\w+):\s*&hex!\(\s*"([0-9A-Fa-f\s]+)"\s*\)', body))
        if fields:
            out.append({k: re.sub(r"\s+", "", v).lower() for k, v in fields.items()})
    return out


def field_vectors(curve):
    path = os.path.join(REF, curve, "src/test_vectors/field.rs")
    if not os.path.exists(path):
        return None
    return hexes(const_block(open(path).read(), "DBL_TEST_VECTORS"))


def schnorr_vectors():
    """BIP340 vectors held by k256/src/schnorr.rs (the public bip-0340/test-vectors.csv data)."""
    text = open(os.path.join(REF, "k256/src/schnorr.rs")).read()
    clean = lambda h: re.sub(r"\s+", "", h).lower()
    field = lambda body, name: clean(re.search(r'%s:\s*hex!\(\s*"([0-9A-Fa-f\s]+)"' % name, body).group(1))
    out = []
    for kind in ("SignVector", "VerifyVector"):
        for body in re.findall(r"%s\s*\{(.*?)\n        \}," % kind, text, re.S):
            if "index:" not in body or "hex!" not in body:
                continue
            v = {"index": int(re.search(r"index:\s*(\d+)", body).group(1)), "public_key": field(body, "public_key"),
                 "message": field(body, "message"), "signature": field(body, "signature")}
            m = re.search(r"valid:\s*(true|false)", body)
            v["valid"] = True if kind == "SignVector" else (m.group(1) == "true")
            out.append(v)
    # index 15-18: one signing key, messages of 0 / 1 / 17 / 100 bytes; the public key is derived by the tests
    sk = clean(re.search(r'SigningKey::from_bytes\(\s*&hex!\("([0-9A-Fa-f]+)"\)', text).group(1))
    ext = text[text.index("let bip340_ext_sign_vectors"):]
    ext = ext[: ext.index("];")]
    for body in re.findall(r"Bip340ExtTest\s*\{(.*?)\n            \}", ext, re.S):
        idx = int(re.search(r"index:\s*(\d+)", body).group(1))
        sig = clean(re.search(r'signature:\s*hex!\(\s*"([0-9A-Fa-f\s]+)"', body).group(1))
        m = re.search(r"msg:\s*(.*?),\n", body, re.S).group(1)
        if m.startswith("vec![]"):
            msg = ""
        elif "hex!" in m:
            msg = clean(re.search(r'hex!\("([0-9A-Fa-f]+)"\)', m).group(1))
        else:                                   # vec![0x99; 100]
            byte, count = re.search(r"vec!\[0x([0-9A-Fa-f]+);\s*(\d+)\]", m).groups()
            msg = byte.lower() * int(count)
        out.append({"index": idx, "secret_key": sk, "message": msg, "signature": sig, "valid": True})
    return sorted(out, key=lambda v: v["index"])


def recovery_vectors():
    """Public-key recovery vectors held by k256/src/ecdsa.rs (mod recovery)."""
    text = open(os.path.join(REF, "k256/src/ecdsa.rs")).read()
    clean = lambda h: re.sub(r"\s+", "", h).lower()
    block = text[text.index("const RECOVERY_TEST_VECTORS"):]
    block = block[: block.index("];")]
    out = []
    for body in re.findall(r"RecoveryTestVector\s*\{(.*?)\n            \}", block, re.S):
        pk = clean(re.search(r'pk:\s*hex!\("([0-9A-Fa-f]+)"\)', body).group(1))
        msg = re.search(r'msg:\s*b"([^"]*)"', body).group(1)
        sig = clean(re.search(r'sig:\s*hex!\(\s*"([0-9A-Fa-f\s]+)"', body).group(1))
        y_odd, x_red = re.search(r"RecoveryId::new\((true|false),\s*(true|false)\)", body).groups()
        out.append({"pk_sec1": pk, "msg_ascii": msg, "hash": "sha256", "sig": sig,
                    "recid": (1 if y_odd == "true" else 0) | (2 if x_red == "true" else 0)})
    eth = text[text.index("fn ethereum_end_to_end_example"):]
    eth = eth[: eth.index("verify_prehash")]
    sk, msg, sig = [clean(h) for h in re.findall(r'hex!\(\s*"([0-9A-Fa-f\s]+)"\s*\)', eth)[:3]]
    recid = int(re.search(r"RecoveryId::from_byte\((\d+)\)", eth).group(1))
    out.append({"secret_key": sk, "msg_hex": msg, "hash": "keccak256", "sig": sig, "recid": recid})
    return out


def main():
    summary = {}
    for curve in ("k256", "p256", "p384", "p224", "p192", "p521"):
        data = {
            "source": "RustCrypto/elliptic-curves %s/src/test_vectors/{group,ecdsa,field}.rs" % curve,
            "group": group_vectors(curve),
            "ecdsa": ecdsa_vectors(curve),
        }
        dbl = field_vectors(curve)
        if dbl is not None:
            data["field_dbl"] = dbl
        if curve == "k256":
            data["schnorr"] = schnorr_vectors()
            print("k256: %d BIP340 vectors (indices %s)" % (len(data["schnorr"]), [v["index"] for v in data["schnorr"]]))
            data["recovery"] = recovery_vectors()
            print("k256: %d public-key recovery vectors" % len(data["recovery"]))
        with open(os.path.join(HERE, "%s.json" % curve), "w") as f:
            json.dump(data, f, indent=1)
            f.write("\n")
        summary[curve] = (len(data["group"]["add"]), len(data["group"]["mul"]), len(data["ecdsa"]),
                          len(dbl) if dbl else 0)
    # bignp256: group vectors only (its signatures are not ECDSA); the hex is the LITTLE-endian wire form (`to_repr`)
    data = {"source": "RustCrypto/elliptic-curves bignp256/src/test_vectors/group.rs (little-endian records)",
            "group": group_vectors("bignp256")}
    with open(os.path.join(HERE, "bign256.json"), "w") as f:
        json.dump(data, f, indent=1)
        f.write("\n")
    summary["bign256"] = (len(data["group"]["add"]), len(data["group"]["mul"]), 0, 0)
    for curve, (a, m, e, d) in summary.items():
        print("%s: %d add, %d mul, %d ecdsa, %d field-dbl vectors" % (curve, a, m, e, d))


if __name__ == "__main__":
    main()
