#!/usr/bin/env python3
"""Developer probe (round 5): dct32_from_tiles_kernel over workgroup size x LDS charged per wave, PAIRED in one process (probe build: X266_FT_WG / X266_FT_LDS per launch)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
fw = fh = 32768
nt = (fw // 16) * (fh // 16)
tc, tp = cd.alloc(nt * 512), cd.alloc(nt * 512)
cd.fill_residual_dev(tc.ptr, nt * 256, 1); cd.fill_residual_dev(tp.ptr, nt * 256, 2)
coef = cd.alloc(fw * fh * 2)
cd.stream_sync()
N = 16
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
cfgs = [(64, 8192), (64, 10240), (64, 12288), (128, 8192), (128, 10240), (128, 12288), (256, 8192), (256, 10240), (256, 5120), (128, 6144)]
for rnd in range(3):
    t = timed(lambda: cd.mem_ceiling_dev(0, tc.ptr, coef.ptr, nt * 256)); cp = nt * 512 / t / 1e9
    row = ["copy %.3f TB/s |" % cp]
    for wg, lds in cfgs:
        os.environ["X266_FT_WG"], os.environ["X266_FT_LDS"] = str(wg), str(lds)
        row.append("%.4f" % timed(lambda: cd.dct32_fwd_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, coef.ptr)))
    print(" ".join(row), flush=True)
print("columns (wg, lds per wave):", cfgs)
