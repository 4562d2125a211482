#!/usr/bin/env python3
"""Developer probe (round 5): how long does a kernel take to reach its steady rate after the chip ran something else?  80 ms of kernel A, then 400 launches of kernel B timed
one by one (HIP events): the SATD batch after the read probe / after the copy stream / after itself, the forward DCT32 after the SATD batch."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n, ns = 1 << 20, 1 << 24
x, z, c = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(ns * 4)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 400
ev = [cd.event_create() for _ in range(N + 1)]
K = {"read probe": lambda: cd.mem_ceiling_dev(3, x.ptr, z.ptr, n * 2048), "copy stream": lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048),
     "SATD batch": lambda: cd.satd8x8_dev(x.ptr, c.ptr, ns), "forward DCT32": lambda: cd.dct32_fwd_dev(x.ptr, z.ptr, n), "idle 200 ms": None}
import time
def run(a, b):
    if K[a] is None:
        cd.stream_sync(); time.sleep(0.2)
    else:
        for _ in range(250): K[a]()
    for i in range(N):
        cd.event_record(ev[i]); K[b]()
    cd.event_record(ev[N]); cd.stream_sync()
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    med = lambda s: statistics.median(s)
    print("%-13s then %-13s: launches 0-9 %.4f  10-29 %.4f  30-59 %.4f  60-99 %.4f  100-199 %.4f  200-399 %.4f ms" % (a, b, med(t[:10]), med(t[10:30]), med(t[30:60]), med(t[60:100]), med(t[100:200]), med(t[200:])), flush=True)
for rnd in range(2):
    for a, b in (("read probe", "SATD batch"), ("copy stream", "SATD batch"), ("SATD batch", "SATD batch"), ("idle 200 ms", "SATD batch"), ("SATD batch", "forward DCT32"), ("copy stream", "forward DCT32"),
                 ("idle 200 ms", "forward DCT32"), ("forward DCT32", "forward DCT32")):
        run(a, b)
