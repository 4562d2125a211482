#!/usr/bin/env python3
"""Developer probe (round 5): the SATD batch LDS-DMA kernel over groups per wave x workgroup size x LDS charge, PAIRED in one process on the same buffers."""
import os, sys, statistics, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
x, y = cd.alloc(n * 128), cd.alloc(n * 4)
cd.fill_residual_dev(x.ptr, n * 64, 0x267); cd.stream_sync()
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
def run(v, g, w, l):
    cd.set_option("satd_variant", v); cd.set_option("satd_groups_per_wave", g); cd.set_option("satd_wg_threads", w); cd.set_option("satd_lds_bytes_per_wave", l)
    return timed(lambda: cd.satd8x8_dev(x.ptr, y.ptr, n))
for rnd in range(2):
    print("read probe %.4f  default %.4f" % (timed(lambda: cd.mem_ceiling_dev(3, x.ptr, y.ptr, n * 128)), run(0, 0, 0, 0)))
    rows = []
    for g, w, l in itertools.product((2, 3, 4, 6, 8), (64, 128, 256), (10240, 12288, 16384)):
        if l * (w // 64) > 65536: continue
        rows.append((run(3, g, w, l), g, w, l))
    rows.sort()
    print("   best: " + "  ".join("%.4f (g%d w%d l%d)" % r for r in rows[:8]))
    print("   default shape row: " + "  ".join("%.4f (g%d w%d l%d)" % r for r in rows if r[1:] == (4, 256, 16384)))
