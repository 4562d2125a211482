#!/usr/bin/env python3
"""Developer probe (round 4): does the 8K frame kernel (33 us, one launch per frame) overlap its ramp and tail with the next frame's
when consecutive frames go to different streams?  Pure launches, no events between them: k streams round-robin, wall clock over F frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
nd, ns = 240 * 135, 960 * 540
R = 6
fin = [(torch.empty(nd * 1024, dtype=torch.int16, device="cuda"), torch.empty(ns * 64, dtype=torch.int16, device="cuda")) for _ in range(R)]
fout = [(torch.empty(nd * 1024, dtype=torch.int16, device="cuda"), torch.empty(ns, dtype=torch.int32, device="cuda")) for _ in range(R)]
for i, (a, b) in enumerate(fin):
    cd.fill_residual_dev(a.data_ptr(), a.numel(), 0x266, i * 1000003)
    cd.fill_residual_dev(b.data_ptr(), b.numel(), 0x267, i * 1000003)
torch.cuda.synchronize()
streams = [cd.stream_create() for _ in range(4)]
def run(k, F):
    for f in range(F):
        a, b = fin[f % R]; c, e = fout[f % R]
        cd.frame_lanes_dev(a.data_ptr(), c.data_ptr(), nd, b.data_ptr(), e.data_ptr(), ns, streams[f % k])
    for s in streams[:k]:
        cd.stream_sync(s)
for rnd in range(3):
    for k in (1, 2, 3, 4):
        run(k, 3000)
        t0 = time.perf_counter(); run(k, 2000); dt = time.perf_counter() - t0
        print("%d stream(s): %.2f us per frame  (%.2f TB/s)" % (k, dt / 2000 * 1e6, (2 * nd * 2048 + ns * 132) / (dt / 2000) / 1e12))
# the two lanes as separate kernels on two streams (DCT32 on one, SATD on the other), frames round-robin
def run_split(F):
    for f in range(F):
        a, b = fin[f % R]; c, e = fout[f % R]
        cd.dct32_fwd_dev(a.data_ptr(), c.data_ptr(), nd, streams[0])
        cd.satd8x8_dev(b.data_ptr(), e.data_ptr(), ns, streams[1])
    cd.stream_sync(streams[0]); cd.stream_sync(streams[1])
run_split(3000)
t0 = time.perf_counter(); run_split(2000); dt = time.perf_counter() - t0
print("two lanes as two kernels on two streams: %.2f us per frame" % (dt / 2000 * 1e6))
