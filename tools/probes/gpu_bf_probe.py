#!/usr/bin/env python3
"""Developer probe: forward DCT32, MFMA kernel vs VALU butterfly variant (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

from x266_amd._lib import OP_SATD8X8
NS = 1 << 24
for rnd in range(2):
    for var, name in ((0, "SATD MFMA (default)"), (2, "SATD VALU butterfly")):
        cd.set_option("satd_variant", var)
        for tpb, lds in (((128, 6144),) if var == 0 else ((64, 8192), (128, 8192), (256, 8192), (64, 16384), (128, 12288))):
            cd.set_option("satd_wg_threads", tpb); cd.set_option("satd_lds_bytes_per_wave", lds)
            cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, NS, 5)
            ms = min(cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, NS, 20) for _ in range(3))
            print("%-20s tpb=%3d lds/wave=%5d: %.4f ms  %.3e blocks/s  %.2f TB/s" % (name, tpb, lds, ms, NS / ms * 1e3, NS * 132 / ms / 1e9), flush=True)
cd.set_option("satd_variant", 0); cd.set_option("satd_wg_threads", 0); cd.set_option("satd_lds_bytes_per_wave", 0)
