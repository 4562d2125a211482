#!/usr/bin/env python3
"""Developer probe (round 5): the read stream over a 2 GiB WINDOW sliding through one 8 GiB allocation (argv[1] = step in MiB): is the
fast / slow level of gpu_placement_sweep.py periodic in the address?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 128
span = int(sys.argv[2]) if len(sys.argv) > 2 else 4096           # MiB of offsets covered
win = int(sys.argv[3]) if len(sys.argv) > 3 else 2048            # window, MiB
big = cd.alloc((span + win + 64) << 20)
y = cd.alloc(64 << 20)
cd.fill_residual_dev(big.ptr, ((span + win) << 20) // 2, 1); cd.stream_sync()
N = 12
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
print("base %012x window %d MiB" % (big.ptr, win))
for off in range(0, span + 1, step):
    t = timed(lambda: cd.mem_ceiling_dev(3, big.ptr + (off << 20), y.ptr, win << 20))
    print("offset %5d MiB : %.4f ms  %.3f TB/s %s" % (off, t, (win << 20) / t / 1e9, "#" * int(((win << 20) / t / 1e9 - 6.5) * 40)), flush=True)
