#!/usr/bin/env python3
"""Developer probe: per-launch duration distribution (HIP events around every launch, 300 back to back) of the inverse family:
DCT32 inverse, small-N inverse, fused forward+inverse.  usage: gpu_inv_dist.py [option=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); cd.set_option(k, int(v))
n = 1 << 20
x = torch.empty(n * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x); r = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266); torch.cuda.synchronize()
N = 300
ev = [cd.event_create() for _ in range(N + 1)]
legs = [("dct32 fwd", 4096, lambda: cd.dct32_fwd_dev(x.data_ptr(), z.data_ptr(), n)),
        ("dct32 inv", 4096, lambda: cd.dct32_inv_dev(x.data_ptr(), z.data_ptr(), n)),
        ("dct2 8x8 fwd", 4096, lambda: cd.transform_fwd_dev(0, 8, x.data_ptr(), z.data_ptr(), n * 16, 0)),
        ("dct2 8x8 inv", 4096, lambda: cd.transform_inv_dev(0, 8, x.data_ptr(), z.data_ptr(), n * 16, 0)),
        ("fused fwd+inv", 6144, lambda: cd.dct32_fwd_inv_dev(x.data_ptr(), z.data_ptr(), r.data_ptr(), n))]
for name, unit, fn in legs:
    for _ in range(30): fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    d = sorted(t)
    print("%-14s min %.4f  p10 %.4f  median %.4f  mean %.4f  p90 %.4f  max %.4f | frac at mean %.3f at median %.3f" % (
        name, d[0], d[N // 10], d[N // 2], sum(d) / N, d[9 * N // 10], d[-1], n * unit / (sum(d) / N) / 8e9, n * unit / d[N // 2] / 8e9), flush=True)
