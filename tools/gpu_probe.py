#!/usr/bin/env python3
"""Developer probe (runs on the GPU box): parity of the three kernels against
the oracle + quick timing sweeps.  Not part of the product or the test-suite."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import x266_amd
from x266_amd._lib import OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8
from _util import Oracle, residual_np, fullrange_np, extremes_np, GOLDEN_DIR

orc = Oracle()
cd = x266_amd.Codec(0)
print(cd.device_info())

def report(name, got, want):
    ok = np.array_equal(got, want)
    bad = np.argwhere(got != want)
    print("%-28s %s  (%d mismatches of %d)" % (name, "OK" if ok else "FAIL", len(bad), got.size))
    if not ok:
        print("   first bad idx", bad[:4].tolist(), "got", got[tuple(bad[0])], "want", want[tuple(bad[0])])
    return ok

g = np.load(os.path.join(GOLDEN_DIR, "dct32_fwd.npz"))
allok = report("dct32 fwd golden", cd.dct32_fwd(g["inputs"]), g["outputs"])
x = np.concatenate([residual_np(3000*1024, 1).reshape(-1,1024), fullrange_np(500*1024, 2).reshape(-1,1024), extremes_np(100*1024,3).reshape(-1,1024)])
f = orc.dct32_fwd(x, threads=8)
allok &= report("dct32 fwd random 3600", cd.dct32_fwd(x), f)
allok &= report("dct32 inv (of fwd)", cd.dct32_inv(f), orc.dct32_inv(f, threads=8))
allok &= report("dct32 inv fullrange", cd.dct32_inv(x), orc.dct32_inv(x, threads=8))
s = np.load(os.path.join(GOLDEN_DIR, "satd8x8.npz"))
allok &= report("satd golden", cd.satd8x8(s["inputs"]), s["outputs"])
d = np.concatenate([residual_np(100001*64, 5).reshape(-1,64), fullrange_np(5000*64, 6).reshape(-1,64)])
allok &= report("satd random 105001", cd.satd8x8(d), orc.satd8x8(d, threads=8))
for n in (1, 2, 3, 31, 33, 63, 65):
    allok &= report("satd ragged n=%d" % n, cd.satd8x8(d[:n]), orc.satd8x8(d[:n]))
    allok &= report("dct ragged n=%d" % n, cd.dct32_fwd(x[:n]), f[:n])
print("ALL PARITY", "OK" if allok else "FAIL")

# ---- timing ------------------------------------------------------------------
N = 1 << 20
din = cd.alloc(N * 2048); dout = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
chk = din.download(np.int16, 4096); print("fill parity", np.array_equal(chk, residual_np(4096, 0x266)))
res = {}
for nt in (1, 0):
    cd.set_option("nontemporal", nt)
    for wg in (2, 4, 6, 8, 12, 16):
        cd.set_option("dct32_wgs_per_cu", wg)
        cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 2)
        ms = cd.time_kernel(OP_DCT32_FWD, din.ptr, dout.ptr, N, 10)
        print("fwd nt=%d wgs/cu=%2d  %.3f ms  %.3e blk/s  %.2f TB/s" % (nt, wg, ms, N/ms*1e3, N*4096/ms*1e3/1e12))
        res["fwd_nt%d_wg%d" % (nt, wg)] = ms
cd.set_option("nontemporal", 1)
for wg in (2, 4, 5, 8):
    cd.set_option("dct32_inv_wgs_per_cu", wg)
    cd.time_kernel(OP_DCT32_INV, din.ptr, dout.ptr, N, 2)
    ms = cd.time_kernel(OP_DCT32_INV, din.ptr, dout.ptr, N, 10)
    print("inv wgs/cu=%2d  %.3f ms  %.3e blk/s  %.2f TB/s" % (wg, ms, N/ms*1e3, N*4096/ms*1e3/1e12))
    res["inv_wg%d" % wg] = ms
NS = 1 << 24
sout = cd.alloc(NS * 4)
for wg in (2, 4, 8, 16):
    cd.set_option("satd_wgs_per_cu", wg)
    cd.time_kernel(OP_SATD8X8, din.ptr, sout.ptr, NS, 2)
    ms = cd.time_kernel(OP_SATD8X8, din.ptr, sout.ptr, NS, 10)
    print("satd wgs/cu=%2d  %.3f ms  %.3e blk/s  %.2f TB/s" % (wg, ms, NS/ms*1e3, NS*132/ms*1e3/1e12))
    res["satd_wg%d" % wg] = ms
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
