#!/usr/bin/env python3
"""Developer probe: full-search timing at 4K (GPU box), HIP events per launch, per tile height.
usage: gpu_me_probe.py [satd|sad|all]   -- every configuration's result is compared with the first one's"""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
from _util import me_frames
what = sys.argv[1] if len(sys.argv) > 1 else "all"
cd = x266_amd.Codec(0)
w, h, rng, pad = 3840, 2160, 64, 64
cur, refp = me_frames(w, h, pad, 2160, mv=(5, -3), noise=4)
dc = torch.from_numpy(cur).cuda(); dr = torch.from_numpy(refp).cuda()
nb = (w // 8) * (h // 8)
best = torch.empty(nb * 2, dtype=torch.int32, device="cuda")
org = dr.data_ptr() + pad * refp.strides[0] + pad
ncand = nb * (2 * rng + 1) ** 2
ev = [cd.event_create() for _ in range(21)]

def timed(fn, reps=20):
    for _ in range(30): fn()                       # clocks
    torch.cuda.synchronize()
    for i in range(reps):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[reps])
    d = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(reps)]
    return statistics.median(d), min(d)

ref = None
if what in ("satd", "all"):
    for tr in (8, 4, 2, 0):
        cd.set_option("me_tile_rows", tr)
        fn = lambda: cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr())
        med, mn = timed(fn)
        res = best.clone()
        if ref is None: ref = res
        print("satd tile_rows=%d: median %.3f ms (min %.3f)  %.3e SATD/s  frac_of_floor %.3f  same_result=%s"
              % (tr, med, mn, ncand / med * 1e3, 1.7554 / med, bool(torch.equal(res, ref))), flush=True)
    cd.set_option("me_tile_rows", 0)
if what in ("sad", "all"):
    ref = None
    for tr in (4, 2, 1, 0):
        cd.set_option("me_tile_rows", tr)
        med, mn = timed(lambda: cd.sad_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr()))
        res = best.clone()
        if ref is None: ref = res
        print("SAD search tile_rows=%d: median %.3f ms (min %.3f)  %.3e SAD/s  frac_of_floor %.3f  same_result=%s"
              % (tr, med, mn, ncand / med * 1e3, 0.8777 / med, bool(torch.equal(res, ref))), flush=True)
    cd.set_option("me_tile_rows", 0)
