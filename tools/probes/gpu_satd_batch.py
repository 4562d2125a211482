#!/usr/bin/env python3
"""Developer probe: the two SATD batch kernels over batch sizes (where the crossover kSatdDmaMinBlocks of x266_device.hpp comes from) and launch
shapes -- "satd_variant" 1 = staged kernel (two-wave workgroups, 2 groups per wave, 6 KiB of LDS charged per wave), 3 = LDS-DMA kernel
(four-wave workgroups, 4 groups per wave, 16 KiB) -- next to this box's read streams (xHipMemCeilingDev).  HIP events per launch.
usage: gpu_satd_batch.py [sizes|shapes]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
N = 100
ev = [cd.event_create() for _ in range(N + 1)]


def timed(fn, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i])
        fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t)


def setc(variant, tpb=0, gpw=0, lds=0):
    cd.set_option("satd_variant", variant)
    cd.set_option("satd_wg_threads", tpb)
    cd.set_option("satd_groups_per_wave", gpw)
    cd.set_option("satd_lds_bytes_per_wave", lds)


nmax = 1 << 24
d = torch.empty(nmax * 64, dtype=torch.int16, device="cuda")
out = torch.empty(nmax, dtype=torch.int32, device="cuda")
scr = torch.empty(nmax * 64 // 512 + 16, dtype=torch.int32, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "sizes"
for rnd in range(2):
    print("# round %d" % rnd)
    for kind, name in ((1, "read stream, one XOR per 2 KiB stored"), (3, "read stream, nothing stored")):
        t = timed(lambda: cd.mem_ceiling_dev(kind, d.data_ptr(), scr.data_ptr(), nmax * 128), 30)
        print("%-40s mean %.4f ms %.3f TB/s" % (name, t[0], nmax * 128 / t[0] / 1e9), flush=True)
    if mode == "sizes":
        for n in (1 << 16, 1 << 18, 518400, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24):
            row = []
            for variant in (1, 3):
                setc(variant)
                t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n), 300 if n < (1 << 22) else 40)
                row.append("variant %d: mean %8.2f us median %8.2f us %.3f TB/s" % (variant, t[0] * 1e3, t[1] * 1e3, n * 132 / t[1] / 1e9))
            print("n = %8d blocks | %s | %s" % (n, row[0], row[1]), flush=True)
    else:
        n = nmax
        for variant, tpb, gpw, lds in ((1, 0, 0, 0), (1, 128, 2, 8192), (1, 64, 2, 6144), (3, 0, 0, 0), (3, 256, 4, 12288), (3, 256, 8, 16384), (3, 128, 4, 16384),
                                       (3, 128, 8, 16384), (3, 128, 8, 24576), (3, 64, 8, 24576), (3, 256, 2, 16384)):
            setc(variant, tpb, gpw, lds)
            t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n), 40)
            print("variant %d tpb %3d groups/wave %d lds/wave %5d : mean %.4f ms median %.4f ms %.3f TB/s = %.3f of 8 TB/s" %
                  (variant, tpb, gpw, lds, t[0], t[1], n * 132 / t[0] / 1e9, n * 132 / t[0] / 8e9), flush=True)
setc(0)
