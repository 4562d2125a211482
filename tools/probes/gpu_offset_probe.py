#!/usr/bin/env python3
"""Developer probe: does the relative placement of the input / output buffers matter? (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD
cd = x266_amd.Codec(0)
N = 1 << 20
SZ = N * 2048
big = cd.alloc(3 * SZ + (64 << 20))
print("base %x" % big.ptr)
cd.fill_residual_dev(big.ptr, N * 1024, 0x266); cd.stream_sync()
def t_fwd(i, o, reps=20):
    cd.time_kernel(OP_DCT32_FWD, i, o, N, 3)
    return min(cd.time_kernel(OP_DCT32_FWD, i, o, N, reps) for _ in range(3))
def t_fi(i, c, r, reps=20):
    for _ in range(3): cd.dct32_fwd_inv_dev(i, c, r, N)
    cd.stream_sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): cd.dct32_fwd_inv_dev(i, c, r, N)
        cd.stream_sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3
for pad in (0, 256, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 16 << 20, (16 << 20) + 8192):
    i = big.ptr; o = big.ptr + SZ + pad; r = big.ptr + 2 * SZ + 2 * pad
    ms = t_fwd(i, o); msf = t_fi(i, o, r)
    print("pad %9d : fwd %.4f ms %.3f TB/s | fused %.4f ms %.3f TB/s" % (pad, ms, N * 4096 / ms / 1e9, msf, N * 6144 / msf / 1e9), flush=True)
# separate allocations, as the tests/bench make them
a = cd.alloc(SZ); b = cd.alloc(SZ); c = cd.alloc(SZ)
print("separate allocs %x %x %x" % (a.ptr, b.ptr, c.ptr))
cd.fill_residual_dev(a.ptr, N * 1024, 0x266); cd.stream_sync()
ms = t_fwd(a.ptr, b.ptr); msf = t_fi(a.ptr, b.ptr, c.ptr)
print("separate : fwd %.4f ms %.3f TB/s | fused %.4f ms %.3f TB/s" % (ms, N * 4096 / ms / 1e9, msf, N * 6144 / msf / 1e9))
import torch
x = torch.empty(N * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x); r = torch.empty_like(x); z2 = torch.empty_like(x)
print("torch %x %x %x %x" % (x.data_ptr(), z.data_ptr(), r.data_ptr(), z2.data_ptr()))
cd.fill_residual_dev(x.data_ptr(), N * 1024, 0x266); cd.stream_sync()
ms = t_fwd(x.data_ptr(), z.data_ptr()); msf = t_fi(x.data_ptr(), z2.data_ptr(), r.data_ptr())
print("torch    : fwd %.4f ms %.3f TB/s | fused %.4f ms %.3f TB/s" % (ms, N * 4096 / ms / 1e9, msf, N * 6144 / msf / 1e9))
