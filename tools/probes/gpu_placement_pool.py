#!/usr/bin/env python3
"""Developer probe (round 5): the headline kernels on buffers carved out of ONE large allocation against separately allocated ones, same process,
alternating; the pool is allocated first / after fragmenting the heap (argv[1] = 1)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
GiB = 1 << 30
frag = len(sys.argv) > 1 and sys.argv[1] == "1"
junk = []
if frag:                                                # a used heap: allocations of mixed sizes, every other one freed again
    tmp = [cd.alloc((3 + 29 * i % 200) << 20) for i in range(60)]
    junk = tmp[::2]
    del tmp
pool = cd.alloc(7 * GiB)
N = 16
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
def legs(x, z, y, tag):
    cd.fill_residual_dev(x, n * 1024, 0x266); cd.stream_sync()
    out = []
    for rep in range(2):
        out.append("copy %.4f fwd %.4f inv %.4f fused %.4f satd %.4f" % (
            timed(lambda: cd.mem_ceiling_dev(0, x, z, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x, z, n)), timed(lambda: cd.dct32_inv_dev(x, z, n)),
            timed(lambda: cd.dct32_fwd_inv_dev(x, z, y, n)), timed(lambda: cd.satd8x8_dev(x, y, 1 << 24))))
    print("%-34s %s" % (tag, " | ".join(out)), flush=True)
for trial in range(5):
    base = pool.ptr + (trial * 200 << 20)
    legs(base, base + 2 * GiB + (2 << 20), base + 4 * GiB + (4 << 20), "pool, offset %4d MiB" % (trial * 200))
    spacer = cd.alloc((1 + 37 * trial) << 20)
    x, z, y = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
    legs(x.ptr, z.ptr, y.ptr, "separate allocations #%d" % trial)
    del x, z, y
    if trial % 2: junk.append(spacer)
