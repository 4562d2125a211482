#!/usr/bin/env python3
"""Developer probe: SATD batch (2^24 blocks) against the LDS charge per wave (= cap on resident waves; the kernel's 96 VGPRs cap them at 20 per CU anyway),
groups per wave and workgroup size.  HIP events, mean and median of 100 launches."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda"); out = torch.empty(n, dtype=torch.int32, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267); torch.cuda.synchronize()
N = 100
ev = [cd.event_create() for _ in range(N + 1)]
def run(label):
    for _ in range(30): cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n)
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    print("%-40s mean %.4f ms frac %.3f   median %.4f frac %.3f" % (label, sum(t) / N, n * 132 / (sum(t) / N) / 8e9, statistics.median(t), n * 132 / statistics.median(t) / 8e9), flush=True)
for lds in (4096, 6144, 8192, 10240, 12288, 16384):
    cd.set_option("satd_lds_bytes_per_wave", lds)
    run("lds/wave %d" % lds)
cd.set_option("satd_lds_bytes_per_wave", 6144)
for gpw in (1, 2, 3, 4):
    cd.set_option("satd_groups_per_wave", gpw)
    run("groups/wave %d" % gpw)
cd.set_option("satd_groups_per_wave", 2)
for tpb in (64, 128, 256):
    cd.set_option("satd_wg_threads", tpb)
    run("threads/workgroup %d" % tpb)
