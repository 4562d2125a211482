#!/usr/bin/env python3
"""Developer probe (round 4): SATD batch, scalar-addressed LDS-DMA body (diag_satd_shape 2 / 4 = 2 / 3 slots), with and without
the waves of a workgroup taking turns over its groups, against the default shape and the box's best read stream."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, x266_amd
cd = x266_amd.Codec(0)


def setc(shape, tpb, gpw, lds, il):
    cd.set_option("diag_satd_shape", shape)
    cd.set_option("satd_wg_threads", tpb)
    cd.set_option("satd_groups_per_wave", gpw)
    cd.set_option("satd_lds_bytes_per_wave", lds)
    cd.set_option("diag_satd_interleave", il)


rs = np.random.RandomState(4)
cd.set_option("adaptive_per_wave", 0)
for nb in (1, 31, 32, 33, 63, 1000, 4099, 65537, 262144 + 17):
    blk = rs.randint(-32768, 32768, size=(nb, 64)).astype(np.int16)
    setc(0, 128, 2, 6144, 0)
    want = cd.satd8x8(blk)
    for shape in (2, 4, 5):
        for tpb in (64, 256):
            for gpw in (1, 2, 3, 8):
                for il in (0, 1):
                    setc(shape, tpb, gpw, 16384, il)
                    assert np.array_equal(cd.satd8x8(blk), want), (nb, shape, tpb, gpw, il)
cd.set_option("adaptive_per_wave", 1)
print("# every LDS-DMA shape equals shape 0 on ragged full-range batches", flush=True)

n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
scr = torch.empty(n * 64 // 512, dtype=torch.int32, device="cuda")
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


setc(0, 128, 2, 6144, 0)
cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n)
torch.cuda.synchronize()
ref = out.clone()
configs = [(0, 128, 2, 6144, 0), (0, 128, 2, 8192, 0)]
for shape in (2, 4):
    depth = {2: 2, 4: 3}[shape]
    for tpb in (128, 256):
        for gpw in (2, 3, 4, 6, 8, 12):
            for lds in (8192, 10240, 12288, 16384, 20480, 32768):
                if lds < depth * 4096 or lds * (tpb // 64) > 65536:
                    continue
                for il in (0, 1):
                    configs.append((shape, tpb, gpw, lds, il))
for rnd in range(2):
    print("# round %d" % rnd)
    t = timed(lambda: cd.mem_ceiling_dev(1, d.data_ptr(), scr.data_ptr(), n * 128))
    print("read stream (xHipMemCeilingDev kind 1) mean %.4f ms %.3f TB/s | median %.4f" % (t[0], n * 128 / t[0] / 1e9, t[1]), flush=True)
    for c in configs:
        setc(*c)
        out.zero_()
        t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n))
        same = bool(torch.equal(out, ref))
        print("satd shape %d tpb %3d gpw %2d lds %5d il %d %s mean %.4f ms %.3f TB/s | median %.4f | min %.4f" % (c + ("" if same else "MISMATCH", t[0], n * 132 / t[0] / 1e9, t[1], t[2])), flush=True)
