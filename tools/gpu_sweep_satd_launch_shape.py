#!/usr/bin/env python3
"""Developer probe: SATD batch launch shape (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 24
din = cd.alloc(N * 128); dout = cd.alloc(N * 4)
cd.fill_residual_dev(din.ptr, N * 64, 0x267); cd.stream_sync()
def t():
    cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, 3)
    return min(cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, 20) for _ in range(4))
for rnd in range(2):
  for tpb in (64, 128, 256):
    cd.set_option("satd_wg_threads", tpb)
    for pad in (0, 4096, 8192, 12288):
        cd.set_option("satd_lds_pad_bytes", pad * (tpb // 64))
        row = "tpb=%3d lds/wave=%5d |" % (tpb, pad)
        for st in (0, 1):
            cd.set_option("satd_lds_stage", st)
            for gpw in (1, 2):
                cd.set_option("satd_groups_per_wave", gpw)
                ms = t(); row += " st%d g%d %.3f ms %.2f TB/s |" % (st, gpw, ms, N*132/ms/1e9)
        print(row, flush=True)
