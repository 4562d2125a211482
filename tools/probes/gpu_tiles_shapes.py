#!/usr/bin/env python3
"""Developer probe (round 5, needs a probe build of xTransformTilesDev that reads X266_TILE_LDS per call): the mixed-class tile kernel (configs[3], one launch) over
tiles per wave x workgroup threads x LDS charged per wave, PAIRED in one process on the same buffers, forward and inverse, next to the copy stream."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20                                   # 32x32 tiles
x, y = cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266)
q = np.arange(n)
cls = cd.alloc(n); cls.upload(np.array([3, 2, 6, 1, 5, 0, 4], np.uint8)[(q + q // 4) % 7])
cd.stream_sync()
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
shapes = [(tpw, wg, lds) for tpw in (1, 2, 3, 4) for wg in (64, 128, 256) for lds in (6144, 8192, 10240, 12288)]
for rnd in range(2):
    t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, y.ptr, n * 2048)); print("copy %.4f ms" % t)
    for inv in (0, 1):
        best = []
        for tpw, wg, lds in shapes:
            cd.set_option("tile_tiles_per_wave", tpw); cd.set_option("dct32_wg_threads", wg); os.environ["X266_TILE_LDS"] = str(lds)
            best.append((timed(lambda: cd.transform_tiles_dev(inv, x.ptr, y.ptr, n, 0, cls.ptr)), tpw, wg, lds))
        ship = [b for b in best if b[1:] == (2, 64, 8192)][0][0]
        best.sort()
        print("%s shipped (2, 64, 8192) %.4f | " % ("inv" if inv else "fwd", ship) + " ".join("%d/%d/%d:%.4f" % (b[1], b[2], b[3], b[0]) for b in best[:8]) + " ... worst %.4f" % best[-1][0], flush=True)
