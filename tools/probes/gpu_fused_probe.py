#!/usr/bin/env python3
"""Developer probe (round 5): the fused forward + inverse DCT32 kernel's variants next to this box's copy stream.
   1. every variant's coefficients and reconstruction against round 4's kernel ("dct32_fwdinv_variant" 1), whole batch, with and
      without the coefficient output;  2. per variant: launch shapes (blocks per wave x workgroup size x LDS charge)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x, z, y = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 30
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=20):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t)
variants = [int(a) for a in sys.argv[1:]] or [1, 2, 0, 3]
# ---- 1. equality with round 4's kernel
ns = 1 << 18
cd.set_option("dct32_fwdinv_variant", 1)
cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, ns); cd.stream_sync()
ref_z, ref_y = z.download(np.int16, ns * 1024), y.download(np.int16, ns * 1024)
for v in variants:
    if v == 1: continue
    cd.set_option("dct32_fwdinv_variant", v)
    for bpw in (1, 2, 3, 4, 5, 8):
        cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
        cd.fill_residual_dev(z.ptr, ns * 1024, 99); cd.fill_residual_dev(y.ptr, ns * 1024, 98)
        cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, ns - 3); cd.stream_sync()
        okz = np.array_equal(z.download(np.int16, (ns - 3) * 1024), ref_z[:(ns - 3) * 1024])
        oky = np.array_equal(y.download(np.int16, (ns - 3) * 1024), ref_y[:(ns - 3) * 1024])
        cd.dct32_fwd_inv_dev(x.ptr, 0, y.ptr, ns - 3); cd.stream_sync()
        oky2 = np.array_equal(y.download(np.int16, (ns - 3) * 1024), ref_y[:(ns - 3) * 1024])
        print("variant %d blocks/wave %d: coef %s recon %s recon-only %s" % (v, bpw, okz, oky, oky2), flush=True)
cd.set_option("dct32_fwdinv_blocks_per_wave", 4)
# ---- 2. timing
for rnd in range(2):
    t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048)); print("copy stream %.4f ms  %.3f TB/s" % (t[0], n * 4096 / t[0] / 1e9), flush=True)
    for v in variants:
        cd.set_option("dct32_fwdinv_variant", v)
        for lds in (0,) if v in (1, 2) else (0, 10240, 16384):
            cd.set_option("dct32_fwdinv_lds_bytes_per_wave", lds)
            for tpb in (64, 128, 256):
                cd.set_option("dct32_wg_threads", tpb)
                row = []
                for bpw in (1, 2, 4, 6, 8, 16):
                    cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
                    t = timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n))
                    row.append("%d:%.4f" % (bpw, t[1]))
                print("variant %d lds %5d wg %3d  median ms by blocks/wave  %s" % (v, lds, tpb, "  ".join(row)), flush=True)
        cd.set_option("dct32_fwdinv_lds_bytes_per_wave", 0)
    # recon only (4096 B/block)
    for v in variants:
        cd.set_option("dct32_fwdinv_variant", v); cd.set_option("dct32_wg_threads", 128); cd.set_option("dct32_fwdinv_blocks_per_wave", 4)
        t = timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, 0, y.ptr, n))
        print("variant %d recon only: %.4f ms (median %.4f)  %.3f TB/s" % (v, t[0], t[1], n * 4096 / t[1] / 1e9), flush=True)
