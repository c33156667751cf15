#!/usr/bin/env python3
"""What the GPU box gives this process in host CPU: affinity, cgroup quota, load, and the oracle's thread scaling measured
with plain C threads-free Python threads (ctypes releases the GIL) at 1, 2, 4, ... threads."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
oracle_lib.build()
rng = np.random.default_rng(1)
n = 4096
s = oracle_lib.scalar_reduce(0, rng.integers(0, 256, n * 32, dtype=np.uint8))
oracle_lib.batch_mul_base(0, s[: 32 * 64])
t0 = time.perf_counter()
oracle_lib.batch_mul_base(0, s)
single = n / (time.perf_counter() - t0)
print("1 thread: %.0f fixed-base k256 mul/s" % single)
for t in (2, 4, 8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    with ThreadPoolExecutor(t) as ex:
        list(ex.map(lambda _: oracle_lib.batch_mul_base(0, s), range(t)))
    dt = time.perf_counter() - t0
    print("%3d threads: %.0f /s  = %.1fx" % (t, t * n / dt, t * n / dt / single), flush=True)
