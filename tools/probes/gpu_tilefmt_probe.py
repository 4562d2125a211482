#!/usr/bin/env python3
"""Developer probe: frame container conversion and residual formation (tile_kernels.hip) on a 32768x32768 luma frame,
HIP events, mean of 30 launches.  The units-per-wave variants were selected through X266_K_* while that switch existed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
w = h = 32768 if len(sys.argv) < 2 else int(sys.argv[1])
nt = (w // 16) * (h // 16)
tiles = torch.randint(0, 256, (nt * 512,), device="cuda", dtype=torch.uint8)
pred = torch.randint(0, 256, (nt * 512,), device="cuda", dtype=torch.uint8)
y = torch.randint(0, 256, (w * h,), device="cuda", dtype=torch.uint8)
u = torch.randint(0, 256, (w * h // 4,), device="cuda", dtype=torch.uint8); v = u.clone()
res = torch.empty(w * h, dtype=torch.int16, device="cuda")
ev = [cd.event_create() for _ in range(2)]
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    cd.event_record(ev[0])
    for _ in range(n): fn()
    cd.event_record(ev[1])
    return cd.event_elapsed_ms(ev[0], ev[1]) / n
legs = (("conv_input_fmt", w * h * 3, lambda: cd.conv_input_fmt_dev(tiles.data_ptr(), y.data_ptr(), u.data_ptr(), v.data_ptr(), w, w, h)),
        ("conv_output_420", w * h * 3, lambda: cd.conv_output_420_dev(tiles.data_ptr(), y.data_ptr(), w, u.data_ptr(), v.data_ptr(), w // 2, w, h)),
        ("residual_luma_32", w * h * 4, lambda: cd.residual_luma_dev(tiles.data_ptr(), pred.data_ptr(), w, h, 32, res.data_ptr())),
        ("residual_luma_8", w * h * 4, lambda: cd.residual_luma_dev(tiles.data_ptr(), pred.data_ptr(), w, h, 8, res.data_ptr())))
for name, nbytes, fn in legs:
    ms = timed(fn)
    print("%-18s %.4f ms  %.2f TB/s  frac %.3f" % (name, ms, nbytes / ms / 1e9, nbytes / ms / 8e9), flush=True)
