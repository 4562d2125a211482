#!/usr/bin/env python3
"""Developer probe (round 5): fine map of the output skew, 0 .. 16 KiB in 256 B steps then 16 .. 256 KiB in 8 KiB steps: copy stream and forward DCT32; separately allocated
buffers (argv[1] = "pool": one allocation)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
N = 10
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=4):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
if len(sys.argv) > 1 and sys.argv[1] == "pool":
    pool = cd.alloc(5 << 30); xp, zp = pool.ptr, pool.ptr + (2 << 30)
else:
    x, z = cd.alloc(n * 2048), cd.alloc(n * 2048 + (1 << 20)); xp, zp = x.ptr, z.ptr
cd.fill_residual_dev(xp, n * 1024, 0x266); cd.stream_sync()
skews = list(range(0, 16384, 256)) + list(range(16384, 262144 + 1, 8192))
res = {}
for rnd in range(2):
    for s in skews:
        res.setdefault(s, []).append((timed(lambda: cd.mem_ceiling_dev(0, xp, zp + s, n * 2048)), timed(lambda: cd.dct32_fwd_dev(xp, zp + s, n))))
for s in skews:
    print("skew %7d : copy %.4f %.4f  fwd %.4f %.4f" % (s, res[s][0][0], res[s][1][0], res[s][0][1], res[s][1][1]))
