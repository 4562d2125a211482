#!/usr/bin/env python3
"""Developer probe (round 5, needs a probe build of launch_intra32_predict that reads X266_IP_UNITS / _ROUNDS / _WG / _PAD per launch): the stand-alone predictor with
short-lived waves (1, 2, 4 or 7 predictions per wave and one round = both dependent fetches exposed per wave) against the shipped long-lived shape (7 x 4 per wave),
paired in one process on the same buffers, over workgroup size and extra LDS charged per workgroup (= fewer resident waves)."""
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
pred = cd.alloc(n * 1024)
cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
def setenv(u, r, wg, pad):
    os.environ.update(X266_IP_UNITS=str(u), X266_IP_ROUNDS=str(r), X266_IP_WG=str(wg), X266_IP_PAD=str(pad))
run = lambda: cd.intra32_predict_dev(refs.ptr, modes.ptr, index.ptr, pred.ptr, n)
for rnd in range(2):
    t = timed(lambda: cd.mem_ceiling_dev(2, pred.ptr, pred.ptr, n * 1024)); print("write stream %.4f ms %.3f TB/s" % (t, n * 1024 / t / 1e9))
    setenv(7, 4, 256, 0); t = timed(run); print("shipped 7 x 4 per wave, wg 256: %.4f ms %.3f TB/s" % (t, n * 1024 / t / 1e9))
    for u, r in ((1, 1), (2, 1), (4, 1), (7, 1), (1, 2), (2, 2), (4, 2), (7, 2), (4, 4), (2, 8)):
        for wg in (64, 256):
            row = []
            for pad in (0, 4096, 8192, 16384, 32768):
                setenv(u, r, wg, pad); t = timed(run); row.append("%5d: %.4f %.2f" % (pad, t, n * 1024 / t / 1e9))
            print("units %d rounds %d wg %3d | pad " % (u, r, wg) + " | ".join(row), flush=True)
