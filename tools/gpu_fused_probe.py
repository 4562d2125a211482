#!/usr/bin/env python3
"""Developer probe (round 4): the fused forward + inverse DCT32 kernel next to this box's copy stream, per launch shape option."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 20
x = torch.empty(n * 1024, dtype=torch.int16, device="cuda")
z = torch.empty_like(x); y = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), x.numel(), 0x266)
torch.cuda.synchronize()
N = 40
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=40):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t)
default_bpw = cd.get_option("dct32_fwdinv_blocks_per_wave")
print("default blocks/wave", default_bpw)
for rnd in range(2):
    t = timed(lambda: cd.mem_ceiling_dev(0, x.data_ptr(), z.data_ptr(), n * 2048)); print("copy stream %.4f ms  %.3f TB/s" % (t[0], n * 4096 / t[0] / 1e9))
    for bpw in (1, 2, 3, 4):
        cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
        for tpb in (64, 128, 256):
            cd.set_option("dct32_wg_threads", tpb)
            t = timed(lambda: cd.dct32_fwd_inv_dev(x.data_ptr(), z.data_ptr(), y.data_ptr(), n))
            print("fused fwd+inv blocks/wave %d wg %3d : %.4f ms (median %.4f)  %.3f TB/s  %.3f of 8 TB/s" % (bpw, tpb, t[0], t[1], n * 6144 / t[0] / 1e9, n * 6144 / t[0] / 8e9))
    cd.set_option("dct32_fwdinv_blocks_per_wave", default_bpw); cd.set_option("dct32_wg_threads", 0)
