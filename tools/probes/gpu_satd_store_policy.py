#!/usr/bin/env python3
"""Developer probe (round 5, needs a probe build of launch_satd8x8 that reads X266_SATD_STORE per launch): the cache policy of the SATD batch's cost stores
(0 nt [shipped] / 1 plain / 2 sc1 nt / 3 sc0 sc1 / 4 sc1 / 5 sc0), PAIRED in one process over several allocation sets -- does a cached cost stream decouple the kernel
from where its 64 MiB output lands?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
ns = 1 << 24
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
keep = []
for aset in range(8):
    x, c = cd.alloc(ns * 128), cd.alloc(ns * 4); keep += [x, c]
    cd.fill_residual_dev(x.ptr, ns * 64, 0x267); cd.stream_sync()
    row = ["read probe %.4f" % timed(lambda: cd.mem_ceiling_dev(3, x.ptr, c.ptr, ns * 128))]
    for rnd in range(2):
        for pol in (0, 1, 2, 3, 4, 5):
            os.environ["X266_SATD_STORE"] = str(pol)
            row.append("%d: %.4f" % (pol, timed(lambda: cd.satd8x8_dev(x.ptr, c.ptr, ns))))
    print("set %d | " % aset + " ".join(row), flush=True)
    keep.append(cd.alloc(((aset * 5) % 7 + 1) * 97 << 20))
