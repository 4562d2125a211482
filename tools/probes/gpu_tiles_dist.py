#!/usr/bin/env python3
"""Developer probe: per-launch duration distribution of the one-launch tile transform (HIP events around every launch), 300 launches back to back."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); cd.set_option(k, int(v))
nt = 1 << 20
x = torch.empty(nt * 1024, dtype=torch.int16, device="cuda"); z = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266); torch.cuda.synchronize()
q = torch.arange(nt, device="cuda")
cls = torch.tensor([3, 2, 6, 1, 5, 0, 4], device="cuda", dtype=torch.uint8)[(q + q // 4) % 7].contiguous()
N = 300
ev = [cd.event_create() for _ in range(N + 1)]
for inv in (0, 1):
    for _ in range(30): cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr())
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); cd.transform_tiles_dev(inv, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr())
    cd.event_record(ev[N])
    d = sorted(cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N))
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    print("inv=%d  min %.4f  p10 %.4f  median %.4f  mean %.4f  p90 %.4f  max %.4f   first ten: %s" % (
        inv, d[0], d[N // 10], d[N // 2], sum(d) / N, d[9 * N // 10], d[-1], " ".join("%.3f" % v for v in t[:10])), flush=True)
    print("        frac at mean %.3f  at median %.3f   mean of launches 100..299: %.4f" % (nt * 4096 / (sum(d) / N) / 8e9, nt * 4096 / d[N // 2] / 8e9, sum(t[100:]) / 200))
# where the slow launches are: indices of launches above 1.08 x median, back to back and with 300 us pauses between launches
import time
for pause in (0.0, 0.0003):
    for i in range(N):
        cd.event_record(ev[i]); cd.transform_tiles_dev(1, x.data_ptr(), z.data_ptr(), nt, 0, cls.data_ptr())
        if pause:
            cd.event_record(ev[N]); torch.cuda.synchronize(); time.sleep(pause)
    cd.event_record(ev[N]) if not pause else None
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N - 1)] if not pause else None
    if t:
        med = statistics.median(t)
        print("back to back: median %.4f, slow launches at" % med, [i for i, v in enumerate(t) if v > 1.08 * med])
