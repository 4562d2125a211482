#!/usr/bin/env python3
"""Developer probe: workgroup size / occupancy cap for the staged DCT32 kernels (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op):
    cd.time_kernel(op, din.ptr, dout.ptr, N, 3)
    return min(cd.time_kernel(op, din.ptr, dout.ptr, N, 20) for _ in range(4))
for tpb in (64, 128, 256):
    cd.set_option("dct32_wg_threads", tpb); cd.set_option("dct32_inv_wg_threads", tpb)
    for per in (2048, 6144, 8192, 10240):
        cd.set_option("dct32_lds_bytes_per_wave", per); cd.set_option("dct32_inv_lds_bytes_per_wave", per)
        row = "tpb=%3d lds/wave=%5d (<=%2d waves/CU) |" % (tpb, per, min(32, 163840 // per))
        for bpw in (1, 2):
            cd.set_option("dct32_blocks_per_wave", bpw); cd.set_option("dct32_inv_blocks_per_wave", bpw)
            f = t(OP_DCT32_FWD); i = t(OP_DCT32_INV)
            row += " bpw%d fwd %.3f ms %.2f TB/s inv %.3f ms %.2f |" % (bpw, f, N*4096/f/1e9, i, N*4096/i/1e9)
        print(row, flush=True)
