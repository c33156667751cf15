#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py::test_generator_table_policy_budget_and_pinning in a fresh process.
The default policy starts a device on the 16-bit table (a 1,024-scalar call does not allocate gigabytes), ECGPU_TABLE_EAGER goes
to the widest at once, a budget caps it, a pinned width (ecgpu_set_base_window) overrides both and 0 un-pins;
ecgpu_base_table_info reports width, bytes and build time; a second context of the device takes the wider table that exists
already.  Results never depend on the width (oracle: the checker)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import oracle_lib
import pyec
from gpu_common import ecgpu_module, rand_scalars

ecgpu = ecgpu_module()
c = pyec.CURVES["p224"]
k = rand_scalars(c.cid, 1024, 0xEC0041F7)
want, winf = oracle_lib.batch_mul_base(c.cid, k)


def check(e):
    out, inf = e.mul_by_generator(c.cid, k)
    assert bytes(out) == bytes(want) and bytes(inf) == bytes(winf)
    return e.base_table_info(c.cid)


a = ecgpu.Engine(0)
assert a.base_table_info(c.cid)["window_bits"] == 0
info = check(a)
assert info["window_bits"] == 16 and 0 < info["bytes"] < (64 << 20) and info["build_ms"] > 0, info
a.set_table_budget(300 << 20)
a.set_table_policy(ecgpu.TABLE_EAGER)
info = check(a)                                           # the widest table within 300 MB
assert 16 < info["window_bits"] < 24 and info["bytes"] <= (300 << 20), info
capped = info["window_bits"]
b = ecgpu.Engine(0)                                       # adaptive, no budget: finds the wider table on the device and uses it
assert check(b)["window_bits"] == capped
a.set_base_window(c.cid, 12)                              # pinned: below any tier
assert check(a)["window_bits"] == 12
a.set_base_window(c.cid, 0)
assert check(a)["window_bits"] == capped
a.set_table_budget(0)
info = check(a)                                           # eager without a budget: the curve's widest
assert info["window_bits"] == 24, info
b.close()
a.close()
print("table policy ok")
