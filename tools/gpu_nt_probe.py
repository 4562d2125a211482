#!/usr/bin/env python3
"""Developer probe: nontemporal load/store bits on the LDS-staged forward kernel and SATD (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op, n, unit, reps=20):
    cd.time_kernel(op, din.ptr, dout.ptr, n, 3)
    ms = min(cd.time_kernel(op, din.ptr, dout.ptr, n, reps) for _ in range(4))
    return ms, n * unit / ms * 1e3 / 1e12
for rnd in range(2):
    for nt in (0, 1, 2, 3, 0):
        cd.set_option("nontemporal", nt)
        print("fwd  nt=%d : %.4f ms %.3f TB/s" % ((nt,) + t(OP_DCT32_FWD, N, 4096)), flush=True)
    for nt, st in ((0, 0), (0, 1), (1, 1), (0, 0)):
        cd.set_option("nontemporal", nt); cd.set_option("satd_lds_stage", st)
        for tpb in (64, 256):
          cd.set_option("satd_wg_threads", tpb)
          for gpw in (1, 2, 4):
            cd.set_option("satd_groups_per_wave", gpw)
            print("satd nt=%d stage=%d tpb=%d gpw=%d : %.4f ms %.3f TB/s" % ((nt, st, tpb, gpw) + t(OP_SATD8X8, 1 << 24, 132)), flush=True)
    cd.set_option("satd_groups_per_wave", 2); cd.set_option("satd_wg_threads", 64); cd.set_option("satd_lds_stage", 0)
