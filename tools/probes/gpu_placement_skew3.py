#!/usr/bin/env python3
"""Developer probe (round 5): bit 12 of (output start - input start), many allocation sets: skew 0 / 4 KiB / 8 KiB (control) / 12 KiB, copy stream, forward and inverse DCT32,
the SATD batch's 64 MiB output, alternating twice per set."""
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
tot = {}
for aset in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    x, z, c = cd.alloc(n * 2048 + (1 << 20)), cd.alloc(n * 2048 + (1 << 20)), cd.alloc((64 << 20) + (1 << 20)); keep += [x, z, c]
    cd.fill_residual_dev(x.ptr, n * 1024, 0x266); cd.stream_sync()
    row = []
    for s in (0, 4096, 8192, 12288, 0, 4096, 8192, 12288):
        t = (timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr + s, n * 2048)), timed(lambda: cd.dct32_fwd_dev(x.ptr, z.ptr + s, n)), timed(lambda: cd.dct32_inv_dev(x.ptr, z.ptr + s, n)),
             timed(lambda: cd.satd8x8_dev(x.ptr, c.ptr + s, 1 << 24)))
        row.append("%5d: %.4f %.4f %.4f %.4f" % ((s,) + t))
        for k, v in zip(("copy", "fwd", "inv", "satd"), t): tot.setdefault((k, s), []).append(v)
    print("set %2d (z - x) mod 2^24 = %#9x | " % (aset, (z.ptr - x.ptr) % (1 << 24)) + " | ".join(row), flush=True)
    keep.append(cd.alloc(((aset * 7) % 5 + 1) * 211 << 20))
for k in ("copy", "fwd", "inv", "satd"):
    print(k, " ".join("%5d: mean %.4f max %.4f" % (s, statistics.mean(tot[(k, s)]), max(tot[(k, s)])) for s in (0, 4096, 8192, 12288)))
