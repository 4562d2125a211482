#!/usr/bin/env python3
"""Developer probe (round 5): on a SLOW landing of the buffers, does another launch shape of the SATD batch / forward kernel recover the speed?  One process, eight
re-allocations, several shapes each."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
shapes = [(0, 0, 0, 0), (1, 0, 0, 0), (3, 2, 256, 16384), (3, 8, 256, 16384), (3, 4, 128, 16384), (3, 4, 256, 12288), (3, 4, 64, 16384), (1, 4, 64, 6144), (1, 2, 256, 6144)]
keep = []
for trial in range(8):
    keep.append(cd.alloc((7 + 53 * trial) << 20))
    x, y = cd.alloc(n * 128), cd.alloc(n * 4)
    cd.fill_residual_dev(x.ptr, n * 64, 0x267); cd.stream_sync()
    row = []
    for variant, gpw, tpb, lds in shapes:
        cd.set_option("satd_variant", variant); cd.set_option("satd_groups_per_wave", gpw); cd.set_option("satd_wg_threads", tpb); cd.set_option("satd_lds_bytes_per_wave", lds)
        row.append("%.4f" % timed(lambda: cd.satd8x8_dev(x.ptr, y.ptr, n)))
    rd = timed(lambda: cd.mem_ceiling_dev(3, x.ptr, y.ptr, n * 128))
    print("read %.4f | " % rd + "  ".join(row), flush=True)
    del x, y
print("columns (variant, groups/wave, wg, lds):", shapes)
