#!/usr/bin/env python3
"""Developer probe (round 4): the few SATD batch candidates that led the sweeps, against each other and the box's read stream, on one more box."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
scr = torch.empty(n * 64, dtype=torch.int16, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267)
torch.cuda.synchronize()
N = 60
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t), min(t)
def setc(shape, tpb, gpw, lds, il):
    cd.set_option("diag_satd_shape", shape); cd.set_option("satd_wg_threads", tpb); cd.set_option("satd_groups_per_wave", gpw)
    cd.set_option("satd_lds_bytes_per_wave", lds); cd.set_option("diag_satd_interleave", il)
configs = [(0, 128, 2, 6144, 0), (2, 256, 3, 12288, 1), (2, 256, 3, 12288, 0), (4, 256, 3, 12288, 1), (2, 256, 8, 16384, 0), (4, 256, 6, 16384, 0), (2, 256, 2, 8192, 1),
           (2, 256, 4, 12288, 1), (2, 256, 4, 16384, 1), (2, 256, 2, 16384, 1), (2, 256, 3, 16384, 1), (2, 128, 3, 12288, 1)]
for rnd in range(3):
    print("# round %d" % rnd)
    for kind, name, nb in ((1, "read", n * 128), (0, "copy", n * 256), (2, "write", n * 128)):
        t = timed(lambda: cd.mem_ceiling_dev(kind, d.data_ptr(), scr.data_ptr(), n * 128))
        print("%-5s stream (xHipMemCeilingDev) mean %.4f ms %.3f TB/s | median %.4f" % (name, t[0], nb / t[0] / 1e9, t[1]), flush=True)
    for c in configs:
        setc(*c)
        t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n))
        print("satd shape %d tpb %3d gpw %2d lds %5d il %d mean %.4f ms %.3f TB/s | median %.4f | min %.4f" % (c + (t[0], n * 132 / t[0] / 1e9, t[1], t[2])), flush=True)
