#!/usr/bin/env python3
"""Developer probe: 32x32 intra prediction throughput (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import x266_amd
from _util import intra_refs_np
cd = x266_amd.Codec(0)
nref = 59918                                             # x 35 modes = 2 097 130 predictions, 2 GiB of output
n = nref * 35
refs = np.zeros((nref, 144), np.uint8); refs[:, :129] = intra_refs_np(nref, 1)
d_r = cd.alloc(refs.nbytes); d_r.upload(refs)
d_p = cd.alloc(n * 1024)
def run(modes, idx, label):
    d_m = cd.alloc(n); d_m.upload(modes.astype(np.uint8))
    d_i = cd.alloc(4 * n); d_i.upload(idx.astype(np.uint32))
    for _ in range(40): cd.intra32_predict_dev(d_r.ptr, d_m.ptr, d_i.ptr, d_p.ptr, n)
    cd.stream_sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20): cd.intra32_predict_dev(d_r.ptr, d_m.ptr, d_i.ptr, d_p.ptr, n)
        cd.stream_sync(); best = min(best, (time.perf_counter() - t0) / 20)
    print("%-28s: %.3f ms  %.3e predictions/s  %.2f TB/s written" % (label, best * 1e3, n / best, n * 1024 / best / 1e12), flush=True)
allm = np.tile(np.arange(35), nref); idx = np.repeat(np.arange(nref), 35)
run(allm, idx, "all 35 modes per border")
for m, name in ((0, "planar"), (1, "DC"), (26, "vertical 26"), (10, "horizontal 10"), (34, "angular 34"), (2, "angular 2"), (21, "angular 21 (neg, vert)"), (15, "angular 15 (neg, horiz)")):
    run(np.full(n, m), idx, name)
