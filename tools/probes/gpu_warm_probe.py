#!/usr/bin/env python3
"""Developer probe: how long does the chip take to reach steady clocks under these kernels? (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
for name, op, n, unit in (("fwd", OP_DCT32_FWD, N, 4096), ("satd", OP_SATD8X8, 1 << 24, 132)):
    time.sleep(2.0)                                            # idle: let the clocks drop
    acc = 0.0
    for i in range(40):
        ms = cd.time_kernel(op, din.ptr, dout.ptr, n, 10)
        acc += ms * 10
        if i < 12 or i % 4 == 0:
            print("%s after %7.1f ms busy: %.4f ms/launch %.3f TB/s" % (name, acc, ms, n * unit / ms / 1e9), flush=True)
