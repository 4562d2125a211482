#!/usr/bin/env python3
"""Developer probe (round 5): xIntra32ResidualDct32Dev (predict -> residual -> DCT32 in one kernel) next to the three-kernel path and this box's streams."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
n_sets = (n + 34) // 35
refs = cd.alloc(n_sets * 144); cd.fill_residual_dev(refs.ptr, n_sets * 72, 3)
modes = cd.alloc(n); modes.upload(np.tile(np.arange(35, dtype=np.uint8), n_sets)[:n])
index = cd.alloc(n * 4); index.upload(np.repeat(np.arange(n_sets, dtype=np.int32), 35)[:n])
src = cd.alloc(n * 1024); cd.fill_residual_dev(src.ptr, n * 512, 4)
coef, pred, res = cd.alloc(n * 2048), cd.alloc(n * 1024), cd.alloc(n * 2048)
cd.stream_sync()
N = 30
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=20):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
for rnd in range(3):
    t = timed(lambda: cd.mem_ceiling_dev(0, res.ptr, coef.ptr, n * 2048)); print("copy stream %.4f ms  %.3f TB/s" % (t, n * 4096 / t / 1e9))
    t = timed(lambda: cd.intra32_residual_dct32_dev(refs.ptr, modes.ptr, index.ptr, src.ptr, coef.ptr, n))
    print("predict -> residual -> DCT32 fused : %.4f ms  %.3e blocks/s  %.3f of 8 TB/s (3072 B/block)" % (t, n / t * 1e3, n * 3072 / t / 8e9))
    t1 = timed(lambda: cd.intra32_predict_dev(refs.ptr, modes.ptr, index.ptr, pred.ptr, n))
    t2 = timed(lambda: cd.dct32_fwd_dev(res.ptr, coef.ptr, n))
    print("predictor alone %.4f ms, forward DCT32 alone %.4f ms (+ a residual kernel in between: >= %.4f ms for the three)" % (t1, t2, t1 + t2 + n * 4096 / 6.7e9))
