#!/usr/bin/env python3
"""Developer probe: nothing but intra predictor launches on the bench's workload (35 modes per reference set, 2 GiB of predictions) and the
copy stream (whose traffic is known: the counter calibration), for rocprofv3 --pmc passes.  usage: intra_only.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n_sets = 59918
g = torch.Generator(device="cuda"); g.manual_seed(0x32)
refs = torch.randint(0, 256, (n_sets, 144), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
modes = torch.arange(35, device="cuda", dtype=torch.uint8).repeat(n_sets)
index = torch.arange(n_sets, device="cuda", dtype=torch.int32).repeat_interleave(35)
pred = torch.empty(n_sets * 35 * 1024, dtype=torch.uint8, device="cuda")
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    cd.intra32_predict_dev(refs.data_ptr(), modes.data_ptr(), index.data_ptr(), pred.data_ptr(), n_sets * 35)
    cd.mem_ceiling_dev(0, src.data_ptr(), pred.data_ptr(), 1 << 30)
torch.cuda.synchronize()
