#!/usr/bin/env python3
"""Developer probe (round 5): input / output buffers inside ONE allocation at spacing D (x at 0, z at D, y at 2 D): the copy stream, the forward and the fused
kernel over D -- is the placement effect a function of the buffers' RELATIVE offset?   argv: first D (MiB), step (MiB), count"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
MiB = 1 << 20
d0, step, count = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (2048, 64, 33)))
pool = cd.alloc((2 * (d0 + step * count) + 2048 + 64) * MiB)
cd.fill_residual_dev(pool.ptr, n * 1024, 0x266); cd.stream_sync()
N = 12
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
for k in range(count):
    D = (d0 + k * step) * MiB
    x, z, y = pool.ptr, pool.ptr + D, pool.ptr + 2 * D
    print("D %6d MiB : copy %.4f  fwd %.4f  fused %.4f  satd %.4f" % (D // MiB, timed(lambda: cd.mem_ceiling_dev(0, x, z, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x, z, n)),
                                                              timed(lambda: cd.dct32_fwd_inv_dev(x, z, y, n)), timed(lambda: cd.satd8x8_dev(x, z, 1 << 24))), flush=True)
