#!/usr/bin/env python3
"""Developer probe: LDS-staged vs direct DCT32 (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV
from _util import Oracle, residual_np, fullrange_np
cd = x266_amd.Codec(0); orc = Oracle()
x = np.concatenate([residual_np(2001 * 1024, 3), fullrange_np(500 * 1024, 4)]).reshape(-1, 1024)
for st in (0, 1):
    cd.set_option("dct32_lds_stage", st)
    f = cd.dct32_fwd(x); print("stage=%d fwd parity" % st, np.array_equal(f, orc.dct32_fwd(x, threads=8)), "inv parity", np.array_equal(cd.dct32_inv(f), orc.dct32_inv(f, threads=8)))
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op):
    cd.time_kernel(op, din.ptr, dout.ptr, N, 3)
    return min(cd.time_kernel(op, din.ptr, dout.ptr, N, 20) for _ in range(4))
for rnd in range(2):
    for st in (0, 1):
        cd.set_option("dct32_lds_stage", st)
        for bpw in (1, 2, 4, 8):
            cd.set_option("dct32_blocks_per_wave", bpw); cd.set_option("dct32_inv_blocks_per_wave", bpw)
            cd.set_option("diag_passthrough", 0); f = t(OP_DCT32_FWD); i = t(OP_DCT32_INV)
            cd.set_option("diag_passthrough", 1); c = t(OP_DCT32_FWD); cd.set_option("diag_passthrough", 0)
            print("stage=%d bpw=%d  fwd %.3f ms %.2f TB/s | inv %.3f ms %.2f TB/s | passthrough %.3f ms %.2f TB/s" % (st, bpw, f, N*4096/f/1e9, i, N*4096/i/1e9, c, N*4096/c/1e9), flush=True)
