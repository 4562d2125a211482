#!/usr/bin/env python3
"""Developer probe: forward DCT32, MFMA kernel vs VALU butterfly variant (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 150)
for rnd in range(2):
    for var, name in ((0, "MFMA (default)"), (2, "VALU butterfly")):
        cd.set_option("dct32_variant", var)
        cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 5)
        ms = min(cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 20) for _ in range(3))
        print("%-16s: %.4f ms  %.3e blocks/s  %.2f TB/s" % (name, ms, N / ms * 1e3, N * 4096 / ms / 1e9), flush=True)
cd.set_option("dct32_variant", 0)
