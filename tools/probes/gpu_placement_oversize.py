#!/usr/bin/env python3
"""Developer probe (round 5): does OVER-allocating each buffer (its own, larger allocation) put it on the fast level?  One process; per trial the buffers are
allocated with `extra` GiB more than they need."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
GiB = 1 << 30
N = 16
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
keep = []
extras = [float(a) for a in sys.argv[1:]] or [0, 1, 2, 6]
for trial in range(2):
    for extra in extras:
        keep.append(cd.alloc((5 + 41 * len(keep)) << 20))
        x, z, y = cd.alloc(int(2 * GiB + extra * GiB)), cd.alloc(int(2 * GiB + extra * GiB)), cd.alloc(int(2 * GiB + extra * GiB))
        cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
        print("extra %.3f GiB: read probe %.4f copy %.4f fwd %.4f fused %.4f satd %.4f" % (
            extra, timed(lambda: cd.mem_ceiling_dev(3, x.ptr, y.ptr, 2 * GiB)), timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048)),
            timed(lambda: cd.dct32_fwd_dev(x.ptr, z.ptr, n)), timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n)), timed(lambda: cd.satd8x8_dev(x.ptr, y.ptr, 1 << 24))), flush=True)
        del x, z, y
