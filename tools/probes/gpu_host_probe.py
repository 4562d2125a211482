#!/usr/bin/env python3
"""Developer probe: PCIe-inclusive rate of the host-pointer batch API (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import x266_amd
cd = x266_amd.Codec(0)
for n in (1 << 17, 1 << 19):
    x = np.random.default_rng(1).integers(-255, 256, size=(n, 1024), dtype=np.int16)
    cd.dct32_fwd(x[:1024])
    import ctypes
    z = np.ones_like(x)                                     # touched: no first-touch page faults inside the timed call
    for rep in range(3):
        t0 = time.perf_counter()
        rc = cd.L.xDct32FwdBatch(cd.ctx, ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(z.ctypes.data), n)
        dt = time.perf_counter() - t0
    print("pageable (touched buffers) n=%7d: %.1f ms  %.3e blocks/s  %.1f GB/s each way" % (n, dt * 1e3, n / dt, n * 2048 / dt / 1e9), flush=True)
    for rep in range(2):
        t0 = time.perf_counter(); z = cd.dct32_fwd(x); dt = time.perf_counter() - t0
    print("pageable (fresh output array) n=%7d: %.1f ms  %.3e blocks/s  %.1f GB/s each way" % (n, dt * 1e3, n / dt, n * 2048 / dt / 1e9), flush=True)
    xp = torch.from_numpy(x).pin_memory(); zp = torch.empty_like(xp).pin_memory()
    for rep in range(3):
        t0 = time.perf_counter()
        rc = cd.L.xDct32FwdBatch(cd.ctx, ctypes.c_void_p(xp.data_ptr()), ctypes.c_void_p(zp.data_ptr()), n)
        dt = time.perf_counter() - t0
    assert rc == 0 and np.array_equal(zp.numpy(), z)
    print("pinned    n=%7d: %.1f ms  %.3e blocks/s  %.1f GB/s each way" % (n, dt * 1e3, n / dt, n * 2048 / dt / 1e9), flush=True)
