#!/usr/bin/env python3
"""Developer probe: cache-policy hints on the LDS-staged kernels (GPU box).
nontemporal bits: 1 nt loads, 2 nt stores, 8 stores use "sc1 nt"."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048); dre = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op, reps=20):
    cd.time_kernel(op, din.ptr, dout.ptr, N, 3)
    ms = min(cd.time_kernel(op, din.ptr, dout.ptr, N, reps) for _ in range(4))
    return ms, N * 4096 / ms * 1e3 / 1e12
def tfi(reps=20):
    for _ in range(3): cd.dct32_fwd_inv_dev(din.ptr, dout.ptr, dre.ptr, N)
    cd.stream_sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): cd.dct32_fwd_inv_dev(din.ptr, dout.ptr, dre.ptr, N)
        cd.stream_sync(); best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3, N * 6144 / best / 1e12
for rnd in range(2):
    for nt in (0, 3, 11, 3, 11):
        cd.set_option("nontemporal", nt)
        print("fwd   nt=%2d : %.4f ms %.3f TB/s" % ((nt,) + t(OP_DCT32_FWD)), flush=True)
    for bpw in (1, 2):
        cd.set_option("dct32_inv_blocks_per_wave", bpw)
        for nt in (3, 11, 3, 11):
            cd.set_option("nontemporal", nt)
            print("inv   nt=%2d bpw=%d : %.4f ms %.3f TB/s" % ((nt, bpw) + t(OP_DCT32_INV)), flush=True)
    cd.set_option("dct32_inv_blocks_per_wave", 2)
    for bpw in (1, 4, 8):
        cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
        for nt in (3, 11):
            cd.set_option("nontemporal", nt)
            print("fused nt=%2d bpw=%d : %.4f ms %.3f TB/s" % ((nt, bpw) + tfi()), flush=True)
    cd.set_option("dct32_fwdinv_blocks_per_wave", 8)
