#!/usr/bin/env python3
"""Developer probe: cap resident waves per CU with unused dynamic LDS (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(op, n):
    cd.time_kernel(op, din.ptr, dout.ptr, n, 3)
    return min(cd.time_kernel(op, din.ptr, dout.ptr, n, 20) for _ in range(4))
# LDS per WG (tpb 64 => one wave per WG): 160 KiB / lds = WGs per CU
for tpb in (64, 256):
    cd.set_option("wg_threads", tpb)
    for waves_per_cu in (32, 24, 20, 16, 12, 10, 8, 6, 4):
        wgs = max(1, waves_per_cu * 64 // tpb)
        lds = 0 if waves_per_cu == 32 else (160 * 1024 // wgs) // 256 * 256
        for k in ("dct32_lds_pad_bytes", "dct32_inv_lds_pad_bytes", "satd_lds_pad_bytes"): cd.set_option(k, lds)
        row = "tpb=%3d waves/cu<=%2d lds=%6d |" % (tpb, waves_per_cu, lds)
        for bpw in (1, 2):
            cd.set_option("dct32_blocks_per_wave", bpw)
            f = t(OP_DCT32_FWD, N); row += " fwd bpw%d %.3f ms %.2f TB/s |" % (bpw, f, N*4096/f/1e9)
        for bpw in (2, 8):
            cd.set_option("dct32_inv_blocks_per_wave", bpw)
            i = t(OP_DCT32_INV, N); row += " inv bpw%d %.3f ms %.2f |" % (bpw, i, N*4096/i/1e9)
        for g in (1, 8):
            cd.set_option("satd_groups_per_wave", g)
            s = t(OP_SATD8X8, 1 << 24); row += " satd g%d %.3f ms %.2f |" % (g, s, (1<<24)*132/s/1e9)
        print(row, flush=True)
