#!/usr/bin/env python3
"""Developer probe: SATD full search at several frame sizes x tile heights (HIP events, median of 20): which tile
height the launcher should pick when a frame (or a stripe of a sharded frame) has few tiles."""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
from _util import me_frames
cd = x266_amd.Codec(0)
ev = [cd.event_create() for _ in range(21)]
rng = 64
for (w, h) in ((3840, 2160), (3840, 1080), (3840, 544), (3840, 272), (3840, 136), (1920, 1080), (1280, 720), (640, 360 // 8 * 8)):
    cur, refp = me_frames(w, h, rng, 7, mv=(5, -3), noise=4)
    dc = torch.from_numpy(cur).cuda(); dr = torch.from_numpy(refp).cuda()
    nb = (w // 8) * (h // 8)
    best = torch.empty(nb * 2, dtype=torch.int32, device="cuda")
    org = dr.data_ptr() + rng * refp.strides[0] + rng
    ref = None
    for tr in (8, 4, 2, 0):
        cd.set_option("me_tile_rows", tr)
        fn = lambda: cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr())
        for _ in range(30): fn()
        torch.cuda.synchronize()
        for i in range(20):
            cd.event_record(ev[i]); fn()
        cd.event_record(ev[20])
        d = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(20)]
        res = best.clone()
        if ref is None: ref = res
        tiles = ((w // 8 + 7) // 8) * ((h // 8 + tr - 1) // tr) if tr else 0            # 0 = the launcher's own choice
        print("%4dx%-4d tile_rows=%d: %5d tiles  median %.3f ms  %.3e SATD/s  same=%s" % (w, h, tr, tiles, statistics.median(d), nb * 129 * 129 / statistics.median(d) * 1e3, bool(torch.equal(res, ref))), flush=True)
cd.set_option("me_tile_rows", 0)
