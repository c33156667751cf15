#!/usr/bin/env python3
"""Extracts the reference's Wycheproof ECDSA vectors from its blobby files into tests/golden/wycheproof.json.

    python tests/golden/extract_wycheproof.py [/root/reference]

Sources (read here, in the build container; /root/reference does not exist on the GPU box):
    k256/src/test_vectors/data/wycheproof.blb          ASN.1 DER signatures   (harness: k256/src/ecdsa.rs:263-384)
    k256/src/test_vectors/data/wycheproof-p1316.blb    IEEE P1363 signatures  (same harness, p1363_sig = true)
    p256|p384|p224|p521/src/test_vectors/data/wycheproof.blb                  (ecdsa_core::new_wycheproof_test!,
                                                        p256/src/ecdsa.rs:166-168, p384/src/ecdsa.rs:184-186,
                                                        p224/src/ecdsa.rs:113-115, p521/src/ecdsa.rs:107-109)

File format (blobby 0.4.0, Cargo.lock:129-130, un-vendored): every number is a git-flavoured VLQ (7 bits per byte,
high bit = "more", value = ((value + 1) << 7) + next); the file is
    <number of blobs> <number of dedup entries> { <len> <bytes> }*  { <code> [<bytes>] }*
where an odd code refers to dedup entry code >> 1 and an even code announces code >> 1 inline bytes.  The harness reads
the blobs five at a time: wx, wy, msg, sig, pass (one byte, 1 = must verify).

Output: {"<set>": {"curve": .., "encoding": "der" | "p1363", "source": .., "strings": [hex ..],
                   "vectors": [[wx, wy, msg, sig, pass], ..]}}   with indices into "strings" (the blobs repeat a lot).
"""
import json
import os
import sys

SETS = [
    ("k256_der", "k256", "der", "k256/src/test_vectors/data/wycheproof.blb"),
    ("k256_p1363", "k256", "p1363", "k256/src/test_vectors/data/wycheproof-p1316.blb"),
    ("p256_der", "p256", "der", "p256/src/test_vectors/data/wycheproof.blb"),
    ("p384_der", "p384", "der", "p384/src/test_vectors/data/wycheproof.blb"),
    ("p224_der", "p224", "der", "p224/src/test_vectors/data/wycheproof.blb"),
    ("p521_der", "p521", "der", "p521/src/test_vectors/data/wycheproof.blb"),
]


def read_vlq(d, pos):
    b = d[pos]
    pos += 1
    val = b & 0x7F
    while b & 0x80:
        b = d[pos]
        pos += 1
        val = ((val + 1) << 7) + (b & 0x7F)
    return val, pos


def parse_blobby(data):
    pos = 0
    total, pos = read_vlq(data, pos)
    ndedup, pos = read_vlq(data, pos)
    dedup = []
    for _ in range(ndedup):
        m, pos = read_vlq(data, pos)
        dedup.append(data[pos:pos + m])
        pos += m
    blobs = []
    while pos < len(data):
        code, pos = read_vlq(data, pos)
        if code & 1:
            blobs.append(dedup[code >> 1])
        else:
            m = code >> 1
            if pos + m > len(data):
                raise ValueError("truncated blob")
            blobs.append(data[pos:pos + m])
            pos += m
    if len(blobs) != total:
        raise ValueError("blob count %d != header %d" % (len(blobs), total))
    return blobs


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = {}
    for name, curve, enc, rel in SETS:
        with open(os.path.join(ref, rel), "rb") as f:
            blobs = parse_blobby(f.read())
        if len(blobs) % 5:
            raise ValueError("%s: %d blobs is not a multiple of 5" % (rel, len(blobs)))
        strings, index, vectors = [], {}, []

        def sid(b):
            h = b.hex()
            if h not in index:
                index[h] = len(strings)
                strings.append(h)
            return index[h]

        for i in range(0, len(blobs), 5):
            wx, wy, msg, sig, ok = blobs[i:i + 5]
            if len(ok) != 1 or ok[0] > 1:
                raise ValueError("%s: vector %d has pass = %r" % (rel, i // 5, ok))
            vectors.append([sid(wx), sid(wy), sid(msg), sid(sig), ok[0]])
        out[name] = {"curve": curve, "encoding": enc, "source": rel, "strings": strings, "vectors": vectors}
        print("%-11s %4d vectors, %3d must verify" % (name, len(vectors), sum(v[4] for v in vectors)))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wycheproof.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
