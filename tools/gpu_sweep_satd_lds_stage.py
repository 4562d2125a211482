#!/usr/bin/env python3
"""Developer probe: LDS-staged vs direct SATD batch (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import x266_amd
from x266_amd._lib import OP_SATD8X8
from _util import Oracle, residual_np, fullrange_np
cd = x266_amd.Codec(0); orc = Oracle()
d = np.concatenate([residual_np(100003 * 64, 3), fullrange_np(5001 * 64, 4)]).reshape(-1, 64)
want = orc.satd8x8(d, threads=8)
for st in (0, 1):
    cd.set_option("satd_lds_stage", st)
    ok = all(np.array_equal(cd.satd8x8(d[:n]), want[:n]) for n in (1, 2, 31, 32, 33, 63, 65, 1000, 105004))
    print("stage=%d parity" % st, ok)
N = 1 << 24
din = cd.alloc(N * 128); dout = cd.alloc(N * 4)
cd.fill_residual_dev(din.ptr, N * 64, 0x267); cd.stream_sync()
def t():
    cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, 3)
    return min(cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, 20) for _ in range(4))
for rnd in range(2):
    for st in (0, 1):
        cd.set_option("satd_lds_stage", st)
        for gpw in (1, 2, 4):
            cd.set_option("satd_groups_per_wave", gpw)
            ms = t()
            print("stage=%d gpw=%d  %.3f ms %.2f TB/s %.3e blk/s" % (st, gpw, ms, N*132/ms/1e9, N/ms*1e3), flush=True)
