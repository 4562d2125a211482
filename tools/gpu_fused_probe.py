#!/usr/bin/env python3
"""Developer probe: fused tiles->coefficients / tiles->costs vs the three-kernel path (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
w, h = 32768, 32768                                   # 262144 DCT blocks, 4.19e6 SATD blocks
nt = (w // 16) * (h // 16)
cur = torch.randint(0, 256, (nt * 512,), dtype=torch.uint8, device="cuda"); pred = torch.randint(0, 256, (nt * 512,), dtype=torch.uint8, device="cuda")
res = torch.empty(w * h, dtype=torch.int16, device="cuda"); coef = torch.empty(w * h, dtype=torch.int16, device="cuda")
cost = torch.empty(w * h // 64, dtype=torch.int32, device="cuda")
def timeit(fn, reps=40):
    for _ in range(60): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps
nb32, nb8 = w * h // 1024, w * h // 64
t_f = timeit(lambda: cd.dct32_fwd_from_tiles_dev(cur.data_ptr(), pred.data_ptr(), w, h, coef.data_ptr()))
t_u = timeit(lambda: (cd.residual_luma_dev(cur.data_ptr(), pred.data_ptr(), w, h, 32, res.data_ptr()), cd.dct32_fwd_dev(res.data_ptr(), coef.data_ptr(), nb32)))
print("DCT32: fused %.3f ms (%.3e blocks/s, %.2f TB/s of 4 KiB/block) | residual+transform %.3f ms (%.3e blocks/s)" % (t_f*1e3, nb32/t_f, nb32*4096/t_f/1e12, t_u*1e3, nb32/t_u))
t_f = timeit(lambda: cd.satd8x8_from_tiles_dev(cur.data_ptr(), pred.data_ptr(), w, h, cost.data_ptr()))
t_u = timeit(lambda: (cd.residual_luma_dev(cur.data_ptr(), pred.data_ptr(), w, h, 8, res.data_ptr()), cd.satd8x8_dev(res.data_ptr(), cost.data_ptr(), nb8)))
print("SATD : fused %.3f ms (%.3e blocks/s, %.2f TB/s of 132 B/block) | residual+cost %.3f ms (%.3e blocks/s)" % (t_f*1e3, nb8/t_f, nb8*132/t_f/1e12, t_u*1e3, nb8/t_u))

for st in (0, 1, 0, 1):
    cd.set_option("satd_lds_stage", st)
    t_f = timeit(lambda: cd.satd8x8_from_tiles_dev(cur.data_ptr(), pred.data_ptr(), w, h, cost.data_ptr()))
    print("SATD from tiles, staged=%d : %.3f ms %.3e blocks/s %.2f TB/s" % (st, t_f * 1e3, nb8 / t_f, nb8 * 132 / t_f / 1e12), flush=True)
for nt_ in (0, 11, 0, 11):
    cd.set_option("nontemporal", nt_)
    t_f = timeit(lambda: cd.dct32_fwd_from_tiles_dev(cur.data_ptr(), pred.data_ptr(), w, h, coef.data_ptr()))
    print("DCT32 from tiles, nt=%d : %.3f ms %.3e blocks/s %.2f TB/s" % (nt_, t_f * 1e3, nb32 / t_f, nb32 * 4096 / t_f / 1e12), flush=True)
