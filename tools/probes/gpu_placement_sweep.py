#!/usr/bin/env python3
"""Developer probe (round 5): the read stream and the SATD batch kernel over WHERE a 2 GiB input lands: a growing pile of kept spacers
(step MiB each, argv[1]) pushes the buffer through device memory; one process."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 1 << 24
N = 16
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
pile = []
for trial in range(trials):
    x, y = cd.alloc(n * 128), cd.alloc(n * 4)
    cd.fill_residual_dev(x.ptr, n * 64, 0x267); cd.stream_sync()
    t_rd = timed(lambda: cd.mem_ceiling_dev(3, x.ptr, y.ptr, n * 128))
    t_satd = timed(lambda: cd.satd8x8_dev(x.ptr, y.ptr, n))
    t_rd2 = timed(lambda: cd.mem_ceiling_dev(3, x.ptr, y.ptr, n * 128))
    print("pile %6d MiB  x %012x : read probe %.4f / %.4f ms (%.3f TB/s)  satd %.4f ms (%.3f of 8 TB/s)" % (trial * step, x.ptr, t_rd, t_rd2, n * 128 / t_rd / 1e9, t_satd, n * 132 / t_satd / 8e9), flush=True)
    del x, y
    pile.append(cd.alloc(step << 20))
