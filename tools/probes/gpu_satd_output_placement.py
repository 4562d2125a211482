#!/usr/bin/env python3
"""Developer probe: SATD batch vs placement of the (small) output buffer relative to the input (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 24
def t(i, o, reps=20):
    cd.time_kernel(OP_SATD8X8, i, o, N, 3)
    ms = min(cd.time_kernel(OP_SATD8X8, i, o, N, reps) for _ in range(3))
    return ms, N * 132 / ms * 1e3 / 1e12
big = cd.alloc((N * 128) + (N * 4) + (1 << 30))
print("base %x" % big.ptr)
cd.fill_residual_dev(big.ptr, N * 64, 0x267); cd.stream_sync()
MiB = 1 << 20
for rnd in range(2):
    for pad in (0, 4096, 65536, MiB, 2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 66 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 512 * MiB + 4096, 960 * MiB):
        o = big.ptr + N * 128 + pad
        print("out = in_end + %9d : %.4f ms %.3f TB/s" % ((pad,) + t(big.ptr, o)), flush=True)
# output BELOW the input
big2 = cd.alloc((N * 128) + (1 << 30))
cd.fill_residual_dev(big2.ptr + (1 << 30), N * 64, 0x267); cd.stream_sync()
for gap in (66 * MiB, 64 * MiB, 128 * MiB, 100 * MiB, 1000 * MiB):
    i = big2.ptr + (1 << 30); o = i - gap
    print("out = in - %9d : %.4f ms %.3f TB/s" % ((gap,) + t(i, o)), flush=True)
