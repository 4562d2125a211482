#!/usr/bin/env python3
"""Developer probe: SATD defaults vs explicit settings, hipMalloc vs torch buffers (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import x266_amd
from x266_amd._lib import OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 24
def t(i, o, reps=20, stream=0):
    cd.time_kernel(OP_SATD8X8, i, o, N, 3, stream)
    ms = min(cd.time_kernel(OP_SATD8X8, i, o, N, reps, stream) for _ in range(4))
    return ms, N * 132 / ms * 1e3 / 1e12
print({k: cd.get_option(k) for k in ("satd_lds_stage", "satd_wg_threads", "satd_groups_per_wave", "satd_lds_bytes_per_wave", "nontemporal", "adaptive_per_wave")})
din = cd.alloc(N * 128); dout = cd.alloc(N * 4)
cd.fill_residual_dev(din.ptr, N * 64, 0x267); cd.stream_sync()
print("hipMalloc buffers, defaults : %.4f ms %.3f TB/s" % t(din.ptr, dout.ptr), flush=True)
d = torch.empty(N * 64, dtype=torch.int16, device="cuda"); s = torch.empty(N, dtype=torch.int32, device="cuda")
cd.fill_residual_dev(d.data_ptr(), N * 64, 0x267); cd.stream_sync()
st = torch.cuda.current_stream().cuda_stream
print("torch buffers, defaults     : %.4f ms %.3f TB/s" % t(d.data_ptr(), s.data_ptr(), stream=st), flush=True)
print("torch in, hipMalloc out     : %.4f ms %.3f TB/s" % t(d.data_ptr(), dout.ptr), flush=True)
print("hipMalloc in, torch out     : %.4f ms %.3f TB/s" % t(din.ptr, s.data_ptr()), flush=True)
cd.set_option("satd_lds_stage", 0); cd.set_option("satd_wg_threads", 64); cd.set_option("satd_groups_per_wave", 1)
print("direct, hipMalloc           : %.4f ms %.3f TB/s" % t(din.ptr, dout.ptr), flush=True)
print("direct, torch               : %.4f ms %.3f TB/s" % t(d.data_ptr(), s.data_ptr()), flush=True)
print("ptrs %x %x %x %x" % (din.ptr, dout.ptr, d.data_ptr(), s.data_ptr()))
