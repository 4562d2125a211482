#!/usr/bin/env python3
"""Developer probe (round 3): where a motion-search workgroup's time goes.  Needs tools/_ab/libx266hip_timing.so
(tools/probes/me_timing_build.sh): the search kernel stamps s_memtime at start, after the window fill, at each wave's last item and
at the end, 12 x 8 bytes per workgroup into the cost-map pointer."""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
from x266_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "tools", "_ab", "libx266hip_timing.so")
from _util import me_frames
cd = x266_amd.Codec(0)
w, h, rng, pad = 3840, 2160, 64, 64
tr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cd.set_option("me_tile_rows", tr)
cur, refp = me_frames(w, h, pad, 2160, mv=(5, -3), noise=4)
dc = torch.from_numpy(cur).cuda(); dr = torch.from_numpy(refp).cuda()
nb = (w // 8) * (h // 8)
best = torch.empty(nb * 2, dtype=torch.int32, device="cuda")
org = dr.data_ptr() + pad * refp.strides[0] + pad
tiles = (w // 64) * ((h // 8 + tr - 1) // tr)
stamps = torch.zeros(tiles * 12, dtype=torch.int64, device="cuda")
for _ in range(30): cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr())
torch.cuda.synchronize()
cd.satd_search_dev(dc.data_ptr(), cur.strides[0], org, refp.strides[0], w, h, rng, best.data_ptr(), stamps.data_ptr())
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(tiles, 12).astype(np.float64)
t0 = s[:, 0].min()
start, fill, ends, end = s[:, 0] - t0, s[:, 1] - s[:, 0], s[:, 2:10] - s[:, 1:2], s[:, 10] - s[:, 0]
span = (s[:, 10].max() - t0)
print("tile_rows %d, %d tiles, kernel span %.0f ticks" % (tr, tiles, span))
def q(a): return "min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print("workgroup total       :", q(end))
print("  window fill         :", q(fill), " = %.1f %% of the workgroup" % (100 * fill.sum() / end.sum()))
print("  items, slowest wave :", q(ends.max(axis=1)))
print("  items, fastest wave :", q(ends.min(axis=1)), " idle at the barrier: %.1f %% of wave time" % (100 * (ends.max(axis=1, keepdims=True) - ends).sum() / (8 * end.sum())))
print("  reduce + exit       :", q(end - fill - ends.max(axis=1)))
hw = stamps.cpu().numpy().reshape(tiles, 12)[:, 11]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xF)      # (xcc, se, sh, cu)
gaps, firsts, lasts = [], [], []
for x in np.unique(xcc):
    t0x = s[xcc == x, 0].min(); t1x = s[xcc == x, 10].max()
    for c in np.unique(cu[xcc == x]):
        sel = np.where(cu == c)[0]
        ev = sorted([(s[i, 0], 1) for i in sel] + [(s[i, 10], -1) for i in sel])
        # time with fewer than 2 resident workgroups on this CU between the XCD's first start and last end
        res, last_t, under = 0, t0x, 0.0
        for t, d in ev:
            if res < 2: under += (t - last_t) * (2 - res) / 2.0
            res += d; last_t = t
        under += (t1x - last_t)
        gaps.append(under / (t1x - t0x))
c0 = cu[0]
sel = np.where(cu == c0)[0]
base = s[sel, 0].min()
print("one CU's workgroups (start, end) in ticks:", " ".join("(%.0f,%.0f)" % (s[i, 0] - base, s[i, 10] - base) for i in sel[np.argsort(s[sel, 0])]))
print("CUs seen: %d; share of CU-time with a workgroup slot empty (XCD first start .. last end): mean %.3f  p90 %.3f  max %.3f" % (len(gaps), np.mean(gaps), np.percentile(gaps, 90), np.max(gaps)))
per_xcd = [(s[xcc == x, 10].max() - s[xcc == x, 0].min()) for x in np.unique(xcc)]
print("XCD spans (ticks):", " ".join("%.0f" % v for v in per_xcd))
