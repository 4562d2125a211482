#!/usr/bin/env python3
"""Developer probe (round 5): the predictor (xIntra32PredictDev) per mode class -- every block the same mode, next to the bench's 35-mode mix and this
box's write stream; and the fused predict -> residual -> DCT32 kernel the same way."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
n_sets = (n + 34) // 35
refs = cd.alloc(n_sets * 144); cd.fill_residual_dev(refs.ptr, n_sets * 72, 3)
modes = cd.alloc(n)
index = cd.alloc(n * 4); index.upload(np.repeat(np.arange(n_sets, dtype=np.int32), 35)[:n])
src = cd.alloc(n * 1024); cd.fill_residual_dev(src.ptr, n * 512, 4)
coef, pred = cd.alloc(n * 2048), cd.alloc(n * 1024)
cd.stream_sync()
N = 30
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=10):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
t = timed(lambda: cd.mem_ceiling_dev(2, pred.ptr, pred.ptr, n * 1024)); print("write stream %.4f ms %.3f TB/s" % (t, n * 1024 / t / 1e9))
cases = [("mix of 35", None), ("planar 0", 0), ("dc 1", 1), ("h+ 2", 2), ("h+ 6", 6), ("h pure 10", 10), ("h- 11", 11), ("h- 14", 14), ("diag 18", 18),
         ("v- 22", 22), ("v- 25", 25), ("v pure 26", 26), ("v+ 30", 30), ("v+ 34", 34)]
for rnd in range(2):
    for name, m in cases:
        modes.upload(np.tile(np.arange(35, dtype=np.uint8), n_sets)[:n] if m is None else np.full(n, m, np.uint8))
        t1 = timed(lambda: cd.intra32_predict_dev(refs.ptr, modes.ptr, index.ptr, pred.ptr, n))
        t2 = timed(lambda: cd.intra32_residual_dct32_dev(refs.ptr, modes.ptr, index.ptr, src.ptr, coef.ptr, n))
        print("%-10s predictor %.4f ms %.3f TB/s written | fused %.4f ms %.3f of 8 TB/s" % (name, t1, n * 1024 / t1 / 1e9, t2, n * 3072 / t2 / 8e9))
