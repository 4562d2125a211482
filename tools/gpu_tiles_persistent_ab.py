#!/usr/bin/env python3
"""Developer probe: mixed-class tile transform, streaming launch (default) against the persistent kernel with the class images in LDS
("tile_variant" = 1), HIP events, median of 20."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n_ctu = 1 << 18
nt = n_ctu * 4
x = torch.empty(nt * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x); z2 = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266); torch.cuda.synchronize()
q = torch.arange(nt, device="cuda")
ev = [cd.event_create() for _ in range(21)]
def timed(fn):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    for i in range(20):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[20])
    return statistics.median(cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(20))
cases = (("7 classes cycling", torch.tensor([3, 2, 6, 1, 5, 0, 4], device="cuda", dtype=torch.uint8)[(q + q // 4) % 7].contiguous()),
         ("all DCT-II 32", torch.full((nt,), 3, device="cuda", dtype=torch.uint8)))
for label, cls in cases:
    for inv in (0, 1):
        cd.set_option("tile_variant", 0)
        ms = timed(lambda: cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr()))
        print("%-18s inv=%d streaming (default)       : %.4f ms frac %.3f" % (label, inv, ms, nt * 4096 / ms / 1e9 / 8), flush=True)
        for wpc in (1, 2, 3):
            cd.set_option("tile_variant", 1); cd.set_option("tile_wgs_per_cu", wpc)
            ms = timed(lambda: cd.transform_tiles_dev(inv, x.data_ptr(), z2.data_ptr(), nt, 0, cls.data_ptr()))
            print("%-18s inv=%d persistent, images in LDS, %d wg/CU: %.4f ms frac %.3f  same=%s" % (label, inv, wpc, ms, nt * 4096 / ms / 1e9 / 8, bool(torch.equal(z, z2))), flush=True)
cd.set_option("tile_variant", 0)
