#!/usr/bin/env python3
"""Developer probe (round 4): SATD batch, LDS-DMA bodies (diag_satd_shape 2 / 4 / 5 = 2 / 3 / 4 slots per wave) with long runs per
wave and few resident waves, against the default shape, same process, alternating rounds."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267)
torch.cuda.synchronize()
N = 50
ev = [cd.event_create() for _ in range(N + 1)]


def timed(fn, warm=25):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i])
        fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t), min(t)


def setc(shape, tpb, gpw, lds):
    cd.set_option("diag_satd_shape", shape)
    cd.set_option("satd_wg_threads", tpb)
    cd.set_option("satd_groups_per_wave", gpw)
    cd.set_option("satd_lds_bytes_per_wave", lds)


setc(0, 128, 2, 6144)
cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n)
torch.cuda.synchronize()
ref = out.clone()
configs = [(0, 128, 2, 6144), (0, 128, 2, 8192)]
for shape in (2, 4, 5):
    depth = {2: 2, 4: 3, 5: 4}[shape]
    for tpb in (64, 128, 256):
        for gpw in (8, 16, 32, 64):
            for lds in (8192, 12288, 16384, 24576, 32768, 65536):
                if lds < depth * 4096 or lds * (tpb // 64) > 65536:
                    continue
                configs.append((shape, tpb, gpw, lds))
for rnd in range(2):
    print("# round %d" % rnd)
    for c in configs:
        setc(*c)
        out.zero_()
        t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n))
        same = bool(torch.equal(out, ref))
        print("satd shape %d tpb %3d gpw %2d lds %5d %s mean %.4f ms %.3f TB/s | median %.4f | min %.4f" % (c + ("" if same else "MISMATCH", t[0], n * 132 / t[0] / 1e9, t[1], t[2])), flush=True)
