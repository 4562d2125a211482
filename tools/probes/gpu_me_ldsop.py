#!/usr/bin/env python3
"""Developer probe (round 6): the motion search with block coefficients from SMEM -> SGPR (shipped) against the LDS-broadcast operand experiment
(a probe build of me_search.hip read X266_ME_LDSOP on every launch), paired in one process on one 4K frame; winners compared."""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
w, h, rng = 3840, 2160, 64
rs = np.random.RandomState(3)
cur = rs.randint(0, 256, (h, w)).astype(np.uint8)
refp = rs.randint(0, 256, (h + 2 * rng, w + 2 * rng)).astype(np.uint8)
dc, dr = cd.alloc(cur.nbytes), cd.alloc(refp.nbytes)
dc.upload(cur); dr.upload(refp)
nb = (w // 8) * (h // 8)
best = [cd.alloc(nb * 8), cd.alloc(nb * 8), cd.alloc(nb * 8)]
org = dr.ptr + rng * refp.shape[1] + rng
N = 12
ev = [cd.event_create() for _ in range(N + 1)]
def run(mode):
    os.environ["X266_ME_LDSOP"] = str(mode)
    fn = lambda: cd.satd_search_dev(dc.ptr, w, org, refp.shape[1], w, h, rng, best[mode].ptr)
    for _ in range(6): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
for r in range(3):
    a, b, c = run(0), run(1), run(2)
    print("SMEM operands %.4f ms | LDS quarters (variant 1) %.4f ms, ratio %.4f | LDS half rows, the shipped loop's pipelining (variant 2) %.4f ms, ratio %.4f" % (a, b, b / a, c, c / a))
r0 = best[0].download(np.uint8, nb * 8)
print("same records:", bool(np.array_equal(r0, best[1].download(np.uint8, nb * 8))), bool(np.array_equal(r0, best[2].download(np.uint8, nb * 8))))
