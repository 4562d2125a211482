#!/usr/bin/env python3
"""Developer probe (round 5): satd8x8_from_tiles / residual_luma bodies next to this box's read and copy streams (32768^2 luma: 2^24 SATD blocks)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
fw = fh = 32768
nt = (fw // 16) * (fh // 16)
tc, tp = cd.alloc(nt * 512), cd.alloc(nt * 512)
cd.fill_residual_dev(tc.ptr, nt * 256, 1); cd.fill_residual_dev(tp.ptr, nt * 256, 2)
cost, res = cd.alloc(fw * fh // 64 * 4), cd.alloc(fw * fh * 2)
cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=10):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
nb = fw * fh // 64
for rnd in range(int(os.environ.get('ROUNDS', '2'))):
    t = timed(lambda: cd.mem_ceiling_dev(1, tc.ptr, res.ptr, nt * 256)); rd = nt * 256 / t / 1e9
    t = timed(lambda: cd.mem_ceiling_dev(0, tc.ptr, res.ptr, nt * 256)); cp = nt * 512 / t / 1e9
    print("read stream %.3f TB/s, copy stream %.3f TB/s" % (rd, cp))
    for v in (1, 3):
        cd.set_option("satd_variant", v)
        t = timed(lambda: cd.satd8x8_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, cost.ptr))
        print("satd8x8_from_tiles variant %d: %.4f ms  %.3f of 8 TB/s  %.3f of read" % (v, t, nb * 132 / t / 8e9, nb * 132 / t / 1e9 / rd))
    cd.set_option("satd_variant", 0)
    for edge in (32, 8):
        t = timed(lambda: cd.residual_luma_dev(tc.ptr, tp.ptr, fw, fh, edge, res.ptr))
        print("residual_luma %2d: %.4f ms  %.3f of 8 TB/s  %.3f of copy" % (edge, t, fw * fh * 4 / t / 8e9, fw * fh * 4 / t / 1e9 / cp))
