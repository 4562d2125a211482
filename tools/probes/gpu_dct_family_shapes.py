#!/usr/bin/env python3
"""Developer probe (round 5): the streaming transform kernels over workgroup size x LDS charged per wave (= resident waves per CU), PAIRED in one process on the same
buffers (probe build: X266_DCT_LDS read per call; the workgroup size is the "dct32_wg_threads" option)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x, y = cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
legs = [("fwd", lambda: cd.dct32_fwd_dev(x.ptr, y.ptr, n)), ("inv", lambda: cd.dct32_inv_dev(x.ptr, y.ptr, n)),
        ("fwd8", lambda: cd.transform_fwd_dev(0, 8, x.ptr, y.ptr, n * 16)), ("inv8", lambda: cd.transform_inv_dev(0, 8, x.ptr, y.ptr, n * 16)),
        ("inv4", lambda: cd.transform_inv_dev(1, 4, x.ptr, y.ptr, n * 64))]
cfgs = [(64, 8192), (64, 6144), (64, 10240), (64, 12288), (128, 8192), (128, 10240), (128, 6144), (128, 5120), (256, 8192), (256, 5120)]
for rnd in range(2):
    t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, y.ptr, n * 2048)); print("copy %.4f ms" % t)
    for name, fn in legs:
        row = []
        for wg, lds in cfgs:
            cd.set_option("dct32_wg_threads", wg); os.environ["X266_DCT_LDS"] = str(lds)
            row.append("%.4f" % timed(fn))
        print("%-5s %s" % (name, " ".join(row)), flush=True)
print("columns (wg, lds per wave):", cfgs)
