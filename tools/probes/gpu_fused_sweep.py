#!/usr/bin/env python3
"""Developer probe (round 5): the DMA-fed fused forward + inverse kernel over pipeline depth / store pairing / LDS charge /
workgroup size / blocks per wave; every variant first checked against round 4's kernel on 2^17 + ragged blocks."""
import os, sys, statistics, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x, z, y = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
variants = [int(a) for a in sys.argv[1:]] or [20, 21, 30, 31, 40, 41, 60, 61]
ns = (1 << 17) - 5
cd.set_option("dct32_fwdinv_variant", 1)
cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, ns); cd.stream_sync()
ref_z, ref_y = z.download(np.int16, ns * 1024), y.download(np.int16, ns * 1024)
for v in variants:
    cd.set_option("dct32_fwdinv_variant", v)
    ok = True
    for bpw in (1, 2, 3, 5, 7, 9, 16):
        cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
        cd.fill_residual_dev(z.ptr, ns * 1024, 99); cd.fill_residual_dev(y.ptr, ns * 1024, 98)
        cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, ns); cd.stream_sync()
        ok &= np.array_equal(z.download(np.int16, ns * 1024), ref_z) and np.array_equal(y.download(np.int16, ns * 1024), ref_y)
        cd.fill_residual_dev(y.ptr, ns * 1024, 98)
        cd.dct32_fwd_inv_dev(x.ptr, 0, y.ptr, ns); cd.stream_sync()
        ok &= np.array_equal(y.download(np.int16, ns * 1024), ref_y)
    print("variant %d bit-identical to round 4's kernel: %s" % (v, ok), flush=True)
t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048)); print("copy stream %.4f ms  %.3f TB/s" % (t, n * 4096 / t / 1e9), flush=True)
cd.set_option("dct32_fwdinv_variant", 1); cd.set_option("dct32_wg_threads", 128); cd.set_option("dct32_fwdinv_blocks_per_wave", 4)
print("round 4's kernel, its default shape: %.4f ms" % timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n)), flush=True)
rows = []
for v in variants:
    cd.set_option("dct32_fwdinv_variant", v)
    for lds, tpb, bpw in itertools.product((0, 12288, 16384, 20480, 32768), (64, 128, 256), (2, 4, 8, 16, 32)):
        if lds and lds < (v // 10 + 1) * 2048: continue
        cd.set_option("dct32_fwdinv_lds_bytes_per_wave", lds); cd.set_option("dct32_wg_threads", tpb); cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
        try:
            t = timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n))
        except Exception as e:
            continue
        rows.append((t, v, lds, tpb, bpw))
        print("variant %2d lds %5d wg %3d blocks/wave %2d : %.4f ms  %.3f of 8 TB/s" % (v, lds, tpb, bpw, t, n * 6144 / t / 8e9), flush=True)
rows.sort()
print("---- best 25")
for t, v, lds, tpb, bpw in rows[:25]:
    print("variant %2d lds %5d wg %3d blocks/wave %2d : %.4f ms  %.3f of 8 TB/s" % (v, lds, tpb, bpw, t, n * 6144 / t / 8e9))
t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048)); print("copy stream %.4f ms  %.3f TB/s" % (t, n * 4096 / t / 1e9), flush=True)
