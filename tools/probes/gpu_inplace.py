#!/usr/bin/env python3
"""Developer probe (round 5): do the streaming kernels run faster IN PLACE (output over the input: the write finds the DRAM row its read just opened)?
Forward / inverse DCT32, the copy stream, SATD unaffected (no matching write).  Paired in one process; the out-of-place legs use two placements of the output."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x, y, z = cd.alloc(n * 2048), cd.alloc(n * 2048), cd.alloc(n * 2048)
cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
for rnd in range(3):
    for name, f in (("copy stream", lambda a, b: cd.mem_ceiling_dev(0, a, b, n * 2048)), ("dct32 fwd", lambda a, b: cd.dct32_fwd_dev(a, b, n)),
                    ("dct32 inv", lambda a, b: cd.dct32_inv_dev(a, b, n)), ("dct 8x8 fwd", lambda a, b: cd.transform_fwd_dev(0, 8, a, b, n * 16))):
        cd.fill_residual_dev(x.ptr, n * 1024, 0x266)
        t_xy = timed(lambda: f(x.ptr, y.ptr)); t_xz = timed(lambda: f(x.ptr, z.ptr)); t_yz = timed(lambda: f(y.ptr, z.ptr))
        t_zz = timed(lambda: f(z.ptr, z.ptr)); t_yy = timed(lambda: f(y.ptr, y.ptr))
        print("%-12s out of place %.4f %.4f %.4f ms | in place %.4f %.4f ms  (%.3f / %.3f of 8 TB/s)" % (name, t_xy, t_xz, t_yz, t_zz, t_yy, n * 4096 / min(t_xy, t_xz, t_yz) / 8e9, n * 4096 / min(t_zz, t_yy) / 8e9), flush=True)
