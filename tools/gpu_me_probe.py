#!/usr/bin/env python3
"""Developer probe: ME search timing at 4K (GPU box)."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
from _util import me_frames
cd = x266_amd.Codec(0)
w, h, rng, pad = 3840, 2160, 64, 64
cur, refp = me_frames(w, h, pad, 2160, mv=(5, -3), noise=4)
dc = torch.from_numpy(cur).cuda(); dr = torch.from_numpy(refp).cuda()
nb = (w // 8) * (h // 8)
best = torch.empty(nb * 2, dtype=torch.int32, device="cuda")
org = dr.data_ptr() + pad * refp.strides[0] + pad
for var, tr, rp in ((1, 2, 1), (2, 2, 2), (2, 4, 1), (2, 4, 2), (2, 4, 3), (2, 4, 2)):
    cd.set_option("me_tile_rows", tr); cd.set_option("me_variant", var); cd.set_option("me_row_pairs", rp)
    cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr()); torch.cuda.synchronize()
    t = time.perf_counter(); reps = 5
    for _ in range(reps): cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    ncand = nb * (2 * rng + 1) ** 2
    print("variant=%d tile_rows=%d row_pairs=%d: %.3f ms/frame  %.3e SATD/s" % (var, tr, rp, dt * 1e3, ncand / dt), flush=True)

for tr in (2, 4):
    cd.set_option("me_tile_rows", tr)
    cd.sad_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr()); torch.cuda.synchronize()
    t = time.perf_counter(); reps = 5
    for _ in range(reps): cd.sad_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print("SAD search tile_rows=%d: %.3f ms/frame  %.3e SAD/s" % (tr, dt * 1e3, ncand / dt), flush=True)
cd.set_option("me_tile_rows", 4)
