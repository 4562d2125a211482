#!/usr/bin/env python3
"""Developer probe (round 4): the 32x32 inverse (and forward) through its own kernel against the same buffer through the one-launch tile kernel
with every tile of class (DCT-II, 32) -- does the tile kernel's structure (two tiles parked in LDS up front) beat the register prefetch?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x = torch.empty(n * 1024, dtype=torch.int16, device="cuda")
z = torch.empty_like(x)
z2 = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266)
cls3 = torch.full((n,), 3, dtype=torch.uint8, device="cuda")
cls_mix = torch.tensor([3, 2, 6, 1, 5, 0, 4], device="cuda", dtype=torch.uint8)[(torch.arange(n, device="cuda") + torch.arange(n, device="cuda") // 4) % 7].contiguous()
torch.cuda.synchronize()
N = 40
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=40):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t)
cd.dct32_inv_dev(x.data_ptr(), z.data_ptr(), n); cd.transform_tiles_dev(1, x.data_ptr(), z2.data_ptr(), n, 0, cls3.data_ptr()); torch.cuda.synchronize()
print("tile kernel == dct32 inverse:", bool(torch.equal(z, z2)))
for rnd in range(3):
    t = timed(lambda: cd.mem_ceiling_dev(0, x.data_ptr(), z.data_ptr(), n * 2048)); print("copy stream      %.4f ms" % t[0])
    for bpw in (1, 2, 3, 4):
        cd.set_option("dct32_inv_blocks_per_wave", bpw)
        t = timed(lambda: cd.dct32_inv_dev(x.data_ptr(), z.data_ptr(), n)); print("dct32 inv  blocks/wave %d : %.4f ms (median %.4f) %.3f of 8 TB/s" % (bpw, t[0], t[1], n * 4096 / t[0] / 8e9))
    cd.set_option("dct32_inv_blocks_per_wave", 2)
    for tpw in (2, 4):
        cd.set_option("tile_tiles_per_wave", tpw)
        t = timed(lambda: cd.transform_tiles_dev(1, x.data_ptr(), z2.data_ptr(), n, 0, cls3.data_ptr())); print("tiles inv, all (DCT-II,32), tiles/wave %d : %.4f ms (median %.4f) %.3f" % (tpw, t[0], t[1], n * 4096 / t[0] / 8e9))
        t = timed(lambda: cd.transform_tiles_dev(1, x.data_ptr(), z2.data_ptr(), n, 0, cls_mix.data_ptr())); print("tiles inv, seven-class mix,  tiles/wave %d : %.4f ms (median %.4f) %.3f" % (tpw, t[0], t[1], n * 4096 / t[0] / 8e9))
    cd.set_option("tile_tiles_per_wave", 0)
    t = timed(lambda: cd.dct32_fwd_dev(x.data_ptr(), z.data_ptr(), n)); print("dct32 fwd : %.4f ms %.3f" % (t[0], n * 4096 / t[0] / 8e9))
    t = timed(lambda: cd.transform_tiles_dev(0, x.data_ptr(), z2.data_ptr(), n, 0, cls3.data_ptr())); print("tiles fwd, all (DCT-II,32): %.4f ms %.3f" % (t[0], n * 4096 / t[0] / 8e9))
    for n4 in (4, 8, 16):
        cnt = n * 1024 // (n4 * n4)
        t = timed(lambda: cd.transform_inv_dev(0, n4, x.data_ptr(), z.data_ptr(), cnt)); print("dct2 %dx%d inv : %.4f ms %.3f" % (n4, n4, t[0], n * 4096 / t[0] / 8e9))
