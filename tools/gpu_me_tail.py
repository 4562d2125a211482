#!/usr/bin/env python3
"""Developer probe: does the motion search pay for partly filled rounds of workgroups?  Frame heights chosen so that the
tile count is 5.00, 5.08 and 5.31 rounds of 768 resident workgroups (3 per CU): time per tile should be flat if not."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
cd = x266_amd.Codec(0)
w, rng, pad = 3840, 64, 64
g = torch.Generator(device="cuda"); g.manual_seed(1)
ev = [cd.event_create() for _ in range(21)]
for h, sp in [(hh, ss) for ss in (1, 2, 3) for hh in (2048, 2080, 2160, 2304, 2464)]:
    cd.set_option("me_splits", sp)
    cur = torch.randint(0, 256, (h, w), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    refp = torch.randint(0, 256, (h + 2 * pad, w + 2 * pad), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    best = torch.empty((h // 8) * (w // 8) * 2, dtype=torch.int32, device="cuda")
    org = refp.data_ptr() + pad * refp.stride(0) + pad
    fn = lambda: cd.satd_search_dev(cur.data_ptr(), cur.stride(0), org, refp.stride(0), w, h, rng, best.data_ptr())
    for _ in range(20): fn()
    torch.cuda.synchronize()
    for i in range(20):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[20])
    d = statistics.median(cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(20))
    tiles = (w // 64) * ((h // 8 + 3) // 4)
    print("splits %d height %4d: %5d tiles = %.2f rounds of 768: %.3f ms, %.4f us per tile" % (sp, h, tiles, tiles / 768.0, d, d * 1e3 / tiles), flush=True)
