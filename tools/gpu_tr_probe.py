#!/usr/bin/env python3
"""Developer probe: transform-set throughput (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
BYTES = 2 << 30
x = torch.empty(BYTES // 2, dtype=torch.int16, device="cuda"); z = torch.empty_like(x)
cd.fill_residual_dev(x.data_ptr(), BYTES // 2, 0x266); torch.cuda.synchronize()
for tpw, st in ((1, 0), (1, 1)):
    cd.set_option("tr_tiles_per_wave", tpw); cd.set_option("tr_lds_stage", st)
    for ttype, n, simple in ((0, 4, 0), (0, 8, 0), (0, 16, 0), (1, 4, 0), (1, 16, 0), (0, 32, 1)):
        cd.set_option("diag_tr32_simple", simple)
        nb = BYTES // (2 * n * n)
        for _ in range(3): cd.transform_fwd_dev(ttype, n, x.data_ptr(), z.data_ptr(), nb)
        torch.cuda.synchronize(); t = time.perf_counter(); reps = 20
        for _ in range(reps): cd.transform_fwd_dev(ttype, n, x.data_ptr(), z.data_ptr(), nb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        print("stage=%d tpw=%d simple=%d type=%d N=%2d: %.3f ms  %.3e blocks/s  %.2f TB/s" % (st, tpw, simple, ttype, n, dt * 1e3, nb / dt, 2 * BYTES / dt / 1e12), flush=True)
