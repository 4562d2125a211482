#!/usr/bin/env python3
"""Developer probe: launch shape of the staged forward / inverse DCT32 kernels under the
nontemporal cache policy (GPU box).  Warm first: the chip needs ~50 ms of load."""
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
    cd.time_kernel(op, din.ptr, dout.ptr, N, 5)
    return min(cd.time_kernel(op, din.ptr, dout.ptr, N, 20) for _ in range(3))
cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 150)
for name, op, tk, lk, bk in (("fwd", OP_DCT32_FWD, "dct32_wg_threads", "dct32_lds_bytes_per_wave", "dct32_blocks_per_wave"),
                             ("inv", OP_DCT32_INV, "dct32_inv_wg_threads", "dct32_inv_lds_bytes_per_wave", "dct32_inv_blocks_per_wave")):
    res = []
    for tpb in (64, 128, 256):
        cd.set_option(tk, tpb)
        for lds in (2048, 4096, 6144, 8192, 10240, 12288, 16384):
            cd.set_option(lk, lds)
            for bpw in (1, 2, 3):
                cd.set_option(bk, bpw)
                ms = t(op); res.append((N * 4096 / ms / 1e9, tpb, lds, bpw))
                print("%s tpb=%3d lds/wave=%5d bpw=%d : %.4f ms %.3f TB/s" % (name, tpb, lds, bpw, ms, res[-1][0]), flush=True)
    print(name, "best:", sorted(res, reverse=True)[:8], flush=True)
