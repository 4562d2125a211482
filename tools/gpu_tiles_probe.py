#!/usr/bin/env python3
"""Developer probe: mixed-class tile transform (xTransformTilesDev) launch shape (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n_ctu = 1 << 18
nt = n_ctu * 4
x = torch.empty(nt * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266); torch.cuda.synchronize()
q = torch.arange(nt, device="cuda")
for label, cls in (("7 classes cycling", torch.tensor([3, 2, 6, 1, 5, 0, 4], device="cuda", dtype=torch.uint8)[(q + q // 4) % 7].contiguous()),
                   ("all DCT-II 32", torch.full((nt,), 3, device="cuda", dtype=torch.uint8)),
                   ("all DCT-II 8", torch.full((nt,), 1, device="cuda", dtype=torch.uint8))):
    for inv in (0, 1):
        tk, lk = ("dct32_inv_wg_threads", "dct32_inv_lds_bytes_per_wave") if inv else ("dct32_wg_threads", "dct32_lds_bytes_per_wave")
        for tpb in (64, 256):
            for lds in (2048, 4096, 8192):
                cd.set_option(tk, tpb); cd.set_option(lk, lds)
                for _ in range(30): cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr())
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr())
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
                print("%-18s inv=%d tpb=%3d lds/wave=%5d: %.4f ms %.2f TB/s" % (label, inv, tpb, lds, dt * 1e3, nt * 4096 / dt / 1e12), flush=True)
        cd.set_option(tk, 64); cd.set_option(lk, 8192)
