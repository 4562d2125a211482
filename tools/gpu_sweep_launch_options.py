#!/usr/bin/env python3
"""Developer probe: sweep launch options of the three kernels (GPU box)."""
import os, sys, itertools, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op, n, unit, reps=10):
    cd.time_kernel(op, din.ptr, dout.ptr, n, 2)
    ms = min(cd.time_kernel(op, din.ptr, dout.ptr, n, reps) for _ in range(3))
    return ms, n * unit / ms * 1e3 / 1e12
which = sys.argv[1:] or ["fwd", "inv", "satd"]
for op, name, n, unit, bpwkey in ((OP_DCT32_FWD, "fwd", N, 4096, "dct32_blocks_per_wave"), (OP_DCT32_INV, "inv", N, 4096, "dct32_blocks_per_wave"), (OP_SATD8X8, "satd", 1 << 24, 132, "satd_groups_per_wave")):
    if name not in which: continue
    vkey = "satd_variant" if name == "satd" else "dct32_variant"
    for nt in (0, 1):
        cd.set_option("nontemporal", nt)
        cd.set_option(vkey, 0)
        for tpb in (64, 256):
            cd.set_option("wg_threads", tpb)
            for bpw in (1, 2, 4, 8, 16):
                cd.set_option(bpwkey, bpw)
                ms, tb = t(op, n, unit)
                print("%-4s stream  nt=%d tpb=%3d per_wave=%2d : %.3f ms %.2f TB/s %.3e/s" % (name, nt, tpb, bpw, ms, tb, n / ms * 1e3), flush=True)
        cd.set_option(vkey, 1); cd.set_option("wg_threads", 256)
        wkey = {"fwd": "dct32_wgs_per_cu", "inv": "dct32_inv_wgs_per_cu", "satd": "satd_wgs_per_cu"}[name]
        for wg in (2, 4, 8):
            cd.set_option(wkey, wg)
            ms, tb = t(op, n, unit)
            print("%-4s persist nt=%d wgs/cu=%d : %.3f ms %.2f TB/s" % (name, nt, wg, ms, tb), flush=True)
    cd.set_option(vkey, 0)
