#!/usr/bin/env python3
"""Developer probe (round 5): the fine skew of gpu_placement_skew.py with SEPARATELY allocated buffers (each over-allocated by 4 MiB; the output starts `skew` bytes into
its allocation): does an odd multiple of 4 KiB help the 1 : 1 streams on every allocation set, and where is the reconstruction-only kernel's optimum?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
N = 12
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=5):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
keep = []
K = 1024
skews = [0, 4 * K, 12 * K, 20 * K, 28 * K, 36 * K, 64 * K, 68 * K, 96 * K, 128 * K, 132 * K, 192 * K, 256 * K, 260 * K, 384 * K, 512 * K, 768 * K, 1024 * K, 1028 * K]
for aset in range(3):
    x, z, y = (cd.alloc(n * 2048 + (4 << 20)) for _ in range(3)); keep += [x, z, y]
    print("allocation set %d: x %#x z %#x y %#x" % (aset, x.ptr, z.ptr, y.ptr))
    cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
    for s in skews:
        print("skew %8d B : copy %.4f  fwd %.4f  inv %.4f  recon-only %.4f  fused(z+s, y+2s) %.4f  fused(z+s, y+s) %.4f  fused(z, y+s) %.4f" % (s,
              timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr + s, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x.ptr, z.ptr + s, n)), timed(lambda: cd.dct32_inv_dev(x.ptr, z.ptr + s, n)),
              timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, 0, y.ptr + s, n)), timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr + s, y.ptr + 2 * s, n)),
              timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr + s, y.ptr + s, n)), timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr + s, n))), flush=True)
    keep.append(cd.alloc((aset + 1) * 333 << 20))
