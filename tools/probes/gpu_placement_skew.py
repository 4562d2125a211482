#!/usr/bin/env python3
"""Developer probe (round 5): input / output buffers inside ONE allocation (one physical landing), the output's start skewed by a FINE offset (256 B .. 1 MiB, and odd
multiples) relative to the input's 2 GiB grid: do the low address bits the streams share (channel / bank interleave) matter for the copy stream, the forward and the fused kernel?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
GiB = 1 << 30
pool = cd.alloc(7 * GiB)
cd.fill_residual_dev(pool.ptr, n * 1024, 0x266); cd.stream_sync()
N = 12
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
skews = [0, 256, 512, 768, 1024, 2048, 3072, 4096, 8192, 12288, 16384, 32768, 65536, 98304, 131072, 262144, 524288, 1 << 20, (1 << 20) + 4096, 3 << 19]
for rnd in range(2):
    for s in skews:
        x, z, y = pool.ptr, pool.ptr + 2 * GiB + s, pool.ptr + 4 * GiB + 2 * s
        print("skew %8d B : copy %.4f  fwd %.4f  inv %.4f  fused %.4f  recon-only %.4f" % (s, timed(lambda: cd.mem_ceiling_dev(0, x, z, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x, z, n)),
              timed(lambda: cd.dct32_inv_dev(x, z, n)), timed(lambda: cd.dct32_fwd_inv_dev(x, z, y, n)), timed(lambda: cd.dct32_fwd_inv_dev(x, 0, y, n))), flush=True)
