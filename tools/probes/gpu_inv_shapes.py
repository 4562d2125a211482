#!/usr/bin/env python3
"""Developer probe (round 5): inverse DCT32 and inverse small transforms over blocks per wave x workgroup size, next to the copy stream and the forward kernel."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x, y = cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=10):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
for rnd in range(2):
    cd.set_option("dct32_wg_threads", 0)
    print("copy %.4f  forward %.4f" % (timed(lambda: cd.mem_ceiling_dev(0, x.ptr, y.ptr, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x.ptr, y.ptr, n))))
    for tpb in (64, 128, 256):
        cd.set_option("dct32_wg_threads", tpb)
        row = []
        for bpw in (1, 2, 3, 4):
            cd.set_option("dct32_inv_blocks_per_wave", bpw)
            row.append("%d: %.4f / %.4f" % (bpw, timed(lambda: cd.dct32_inv_dev(x.ptr, y.ptr, n)), timed(lambda: cd.transform_inv_dev(0, 8, x.ptr, y.ptr, n * 16))))
        print("wg %3d  blocks (tiles) per wave: inverse DCT32 / inverse DCT-II 8x8 (ms)   %s" % (tpb, "   ".join(row)), flush=True)
    cd.set_option("dct32_inv_blocks_per_wave", 2)
