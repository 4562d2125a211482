#!/usr/bin/env python3
"""Developer probe: SATD batch kernel, direct vs LDS-staged (line-dense, nontemporal loads) (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 24
din = cd.alloc(N * 128); dout = cd.alloc(N * 4)
cd.fill_residual_dev(din.ptr, N * 64, 0x267); cd.stream_sync()
def t(reps=20):
    cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, 3)
    ms = min(cd.time_kernel(OP_SATD8X8, din.ptr, dout.ptr, N, reps) for _ in range(4))
    return ms, N * 132 / ms * 1e3 / 1e12
cd.set_option("satd_lds_stage", 0); cd.set_option("nontemporal", 11)
print("direct default : %.4f ms %.3f TB/s" % t(), flush=True)
res = []
for nt in (1, 2, 1, 2):
    cd.set_option("satd_lds_stage", nt)
    for tpb in (64, 128):
        cd.set_option("satd_wg_threads", tpb)
        for lds in (4096, 6144, 8192):
            cd.set_option("satd_lds_bytes_per_wave", lds)
            for gpw in (2, 3, 4):
                cd.set_option("satd_groups_per_wave", gpw)
                ms, tb = t(10)
                res.append((tb, nt, tpb, lds, gpw))
                print("staged stage=%d tpb=%3d lds=%5d gpw=%d : %.4f ms %.3f TB/s" % (nt, tpb, lds, gpw, ms, tb), flush=True)
print("best:", sorted(res, reverse=True)[:6])
cd.set_option("satd_lds_stage", 0); cd.set_option("satd_wg_threads", 64); cd.set_option("satd_groups_per_wave", 1)
print("direct default : %.4f ms %.3f TB/s" % t(), flush=True)
