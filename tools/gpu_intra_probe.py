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
for rounds in (1, 2, 3, 4, 6):
    cd.set_option("intra_rounds", rounds)
    run(allm, idx, "all 35 modes, rounds=%d" % rounds)
cd.set_option("intra_rounds", 4)
for m, name in ((0, "planar"), (1, "DC"), (26, "vertical 26"), (10, "horizontal 10"), (34, "angular 34"), (2, "angular 2"), (21, "angular 21 (neg, vert)"), (15, "angular 15 (neg, horiz)")):
    run(np.full(n, m), idx, name)

# ---- mode decision: all 35 modes of every block, costs only
nb = 1 << 17
refs2 = np.zeros((nb, 144), np.uint8); refs2[:, :129] = intra_refs_np(nb, 2)
src = (np.arange(nb * 1024, dtype=np.uint32) * 2654435761 >> 24).astype(np.uint8)
d_r2 = cd.alloc(refs2.nbytes); d_r2.upload(refs2)
d_s = cd.alloc(nb * 1024); d_s.upload(src)
d_c = cd.alloc(nb * 35 * 4); d_b = cd.alloc(nb)
for _ in range(10): cd.intra32_costs_dev(d_r2.ptr, d_s.ptr, d_c.ptr, d_b.ptr, nb)
cd.stream_sync(); best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): cd.intra32_costs_dev(d_r2.ptr, d_s.ptr, d_c.ptr, d_b.ptr, nb)
    cd.stream_sync(); best = min(best, (time.perf_counter() - t0) / 10)
print("mode decision (35 modes x 16 SATD): %.3f ms per %d blocks  %.3e blocks/s  %.3e mode evaluations/s  %.3e 8x8 SATD/s" % (
    best * 1e3, nb, nb / best, nb * 35 / best, nb * 35 * 16 / best), flush=True)
