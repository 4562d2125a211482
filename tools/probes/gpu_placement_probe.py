#!/usr/bin/env python3
"""Developer probe (round 5): is the +-5 % by which the same kernel moves from process to process a matter of WHERE its buffers land?
One process, the same kernels, the buffers freed and allocated again (with a spacer of varying size in between) eight times."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=10):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
keep = []
for trial in range(8):
    spacer = cd.alloc((1 + 37 * trial) << 20)          # shifts where the next allocations land
    x, z, y = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
    cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
    row = ["x %012x z %012x y %012x" % (x.ptr, z.ptr, y.ptr)]
    for rep in range(2):
        t_copy = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048))
        t_fwd = timed(lambda: cd.dct32_fwd_dev(x.ptr, z.ptr, n))
        t_fus = timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n))
        t_satd = timed(lambda: cd.satd8x8_dev(x.ptr, y.ptr, 1 << 24))
        row.append("copy %.4f fwd %.4f fused %.4f satd %.4f" % (t_copy, t_fwd, t_fus, t_satd))
    print(" | ".join(row), flush=True)
    if trial % 2: keep.append(spacer)                   # some spacers stay allocated: the heap's layout differs from trial to trial
    del x, z, y
