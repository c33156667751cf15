#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 .s file: python tools/isa_hist.py file.s [kernel-substring]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filt not in name: continue
    ops = [l.split()[0] for l in body.splitlines() if re.match(r"^\s+[vs]_|^\s+(global|ds|buffer|scratch|flat)_", l)]
    c = collections.Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print("== %s: %d instr, %d VALU" % (name, len(ops), valu))
    print("   " + ", ".join("%s:%d" % kv for kv in c.most_common(14)))
