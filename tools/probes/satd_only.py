#!/usr/bin/env python3
"""Developer probe: nothing but SATD batch launches (2^24 blocks), for rocprofv3 --pmc passes.  usage: satd_only.py <satd_variant> [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
cd.set_option("satd_variant", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n)
    cd.mem_ceiling_dev(3, d.data_ptr(), out.data_ptr(), n * 128)
torch.cuda.synchronize()
