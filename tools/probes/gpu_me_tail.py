#!/usr/bin/env python3
"""Developer probe (round 3): what a partly filled last round of workgroups costs the motion search.  8-row tiles,
512 resident workgroups (2 per CU): frame heights from 1 to 8.5 rounds; a fixed tail shows as the intercept of
time against tile count."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import x266_amd
cd = x266_amd.Codec(0)
w, rng, pad = 3840, 64, 64
tr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cd.set_option("me_tile_rows", tr)
g = torch.Generator(device="cuda"); g.manual_seed(1)
ev = [cd.event_create() for _ in range(21)]
per_cu = 2
for h in (512, 544, 1024, 1088, 2048, 2112, 2160, 2560, 4096, 4352):
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
    tiles = (w // 64) * ((h // 8 + tr - 1) // tr)
    cand = (h // 8) * (w // 8) * 129 * 129
    print("tile_rows %d height %4d: %5d tiles = %.2f rounds of %d: %.3f ms, %.4f us per tile, frac_of_floor %.3f" % (
        tr, h, tiles, tiles / (256.0 * per_cu), 256 * per_cu, d, d * 1e3 / tiles, cand * 32 / 64 * 4 / (1024 * 2.4e9) * 1e3 / d), flush=True)
