#!/usr/bin/env python3
"""Runs the HBM gather probe (ecgpu_valu_probe(200)): 2^20 lanes x 16 random 64-byte reads of the k256 comb table =
exactly 1 GiB.  Under `rocprofv3 --pmc FETCH_SIZE` the counter of k_gather_probe calibrates FETCH_SIZE for the access
pattern of k_fixed_base (the guide calibrates wide coalesced streams only)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ec = importlib.import_module("elliptic-curves_amd")
e = ec.Engine(0)
if len(sys.argv) > 1:
    e.set_base_window(0, int(sys.argv[1]))
bps = e.valu_probe(200)
print("gather probe: %.1f GB/s of random 64-byte reads (%d bytes per launch)" % (bps / 1e9, (1 << 20) * 16 * 64))
