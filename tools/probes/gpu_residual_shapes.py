#!/usr/bin/env python3
"""Developer probe (round 5): residual_luma_kernel over workgroup size x LDS charge, PAIRED in one process on the same buffers (the probe build of the launcher reads
X266_RES_WG / X266_RES_LDS on every launch)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
fw = fh = 32768
nt = (fw // 16) * (fh // 16)
tc, tp = cd.alloc(nt * 512), cd.alloc(nt * 512)
cd.fill_residual_dev(tc.ptr, nt * 256, 1); cd.fill_residual_dev(tp.ptr, nt * 256, 2)
res = cd.alloc(fw * fh * 2)
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
cfgs = [(256, 0), (128, 8192), (128, 12288), (128, 16384), (128, 20480), (192, 12288), (256, 12288), (256, 24576), (64, 0)]
for rnd in range(3):
    t = timed(lambda: cd.mem_ceiling_dev(0, tc.ptr, res.ptr, nt * 256)); cp = nt * 512 / t / 1e9
    row = ["copy %.3f TB/s |" % cp]
    for wg, lds in cfgs:
        os.environ["X266_RES_WG"], os.environ["X266_RES_LDS"] = str(wg), str(lds)
        a = timed(lambda: cd.residual_luma_dev(tc.ptr, tp.ptr, fw, fh, 32, res.ptr))
        b = timed(lambda: cd.residual_luma_dev(tc.ptr, tp.ptr, fw, fh, 8, res.ptr))
        row.append("%.4f/%.4f" % (a, b))
    print(" ".join(row), flush=True)
print("columns (wg, dynamic lds): 32x32 order / 8x8 order ms", cfgs)
