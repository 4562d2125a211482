#!/usr/bin/env python3
"""Developer probe: arithmetic vs pass-through at the same launch shape (GPU box)."""
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
    return min(cd.time_kernel(op, din.ptr, dout.ptr, N, 20) for _ in range(5))
for rnd in range(2):
  for tpb in (64, 256):
    cd.set_option("wg_threads", tpb)
    for bpw in (1, 2, 3, 4, 6, 8):
        cd.set_option("dct32_blocks_per_wave", bpw); cd.set_option("dct32_inv_blocks_per_wave", bpw)
        cd.set_option("diag_passthrough", 0); f = t(OP_DCT32_FWD); i = t(OP_DCT32_INV)
        cd.set_option("diag_passthrough", 1); c = t(OP_DCT32_FWD)
        print("tpb=%3d bpw=%d  fwd %.3f ms %.2f TB/s | inv %.3f ms %.2f TB/s | passthrough %.3f ms %.2f TB/s" % (tpb, bpw, f, N*4096/f/1e9, i, N*4096/i/1e9, c, N*4096/c/1e9), flush=True)
