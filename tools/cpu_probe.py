#!/usr/bin/env python3
"""How many host CPUs does the GPU box really give this container?  (cgroup quota vs nproc vs affinity)
and how the reference C path scales with pinned threads.  Checker-side tool (uses oracle/_ref)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_cpu as bench
from _util import Oracle, Reference, ref_path

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node' | head -12")
orc = Oracle()
ref = Reference() if os.path.exists(ref_path()) else None
n = 1 << 18
x = orc.fill_residual(n * 1024, 0x266).reshape(n, 1024)
def work(i, o, c):
    if ref: ref.lib.ref_dct32_fwd(ctypes.c_void_p(i.ctypes.data), ctypes.c_void_p(o.ctypes.data), ctypes.c_ulong(c))
    else: orc.lib.orc_dct32_fwd_mt(ctypes.c_void_p(i.ctypes.data), ctypes.c_void_p(o.ctypes.data), ctypes.c_size_t(c), 1)
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if t > os.cpu_count(): break
    dt, _, _ = bench.run_pinned(n, t, work, lambda b, e: (x[b:e].copy(), np.zeros_like(x[b:e])))
    print("threads %3d: %.3e blocks/s  (%.2f x per thread of 1-thread rate)" % (t, n / dt, 0))
