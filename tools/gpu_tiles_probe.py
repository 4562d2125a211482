#!/usr/bin/env python3
"""Developer probe: mixed-class tile transform (xTransformTilesDev) launch shape (GPU box), HIP events, median."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n_ctu = 1 << 18
nt = n_ctu * 4
x = torch.empty(nt * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x)
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
        tk = "dct32_inv_wg_threads" if inv else "dct32_wg_threads"
        shapes = [tuple(int(u) for u in v.split(":")) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [(64, 6144), (64, 8192), (64, 10240), (128, 8192), (256, 8192)]   # threads per workgroup : LDS bytes per wave
        for tpb, lds in shapes:
            for tpw in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4)):
                cd.set_option(tk, tpb); cd.set_option("tile_tiles_per_wave", tpw); cd.set_option("tile_lds_bytes_per_wave", lds)
                ms = timed(lambda: cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr()))
                print("%-18s inv=%d tpb=%3d lds/wave=%d tiles/wave=%d: %.4f ms %.2f TB/s frac %.3f" % (label, inv, tpb, lds, tpw, ms, nt * 4096 / ms / 1e9, nt * 4096 / ms / 1e9 / 8), flush=True)
        cd.set_option(tk, 64)
cd.set_option("tile_tiles_per_wave", 0)
