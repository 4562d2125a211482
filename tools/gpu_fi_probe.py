#!/usr/bin/env python3
"""Developer probe: launch shape of the fused forward+inverse DCT32 kernel (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
N = 1 << 20
din = cd.alloc(N * 2048); dco = cd.alloc(N * 2048); dre = cd.alloc(N * 2048)
cd.fill_residual_dev(din.ptr, N * 1024, 0x266); cd.stream_sync()
def t(reps=20):
    for _ in range(3): cd.dct32_fwd_inv_dev(din.ptr, dco.ptr, dre.ptr, N)
    cd.stream_sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): cd.dct32_fwd_inv_dev(din.ptr, dco.ptr, dre.ptr, N)
        cd.stream_sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3, N * 6144 / best / 1e12
for nt in (3,):
    cd.set_option("nontemporal", nt)
    for tpb in (64, 256):
        cd.set_option("dct32_inv_wg_threads", tpb)
        for lds in (2048, 8192):
            cd.set_option("dct32_inv_lds_bytes_per_wave", lds)
            for bpw in (3, 4, 6, 8, 16):
                cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
                print("nt=%d tpb=%3d lds=%5d bpw=%d : %.4f ms %.3f TB/s" % ((nt, tpb, lds, bpw) + t()), flush=True)
cd2 = x266_amd.Codec(0)
cd = cd2
print("defaults:", {k: cd.get_option(k) for k in ("nontemporal", "dct32_inv_wg_threads", "dct32_inv_lds_bytes_per_wave", "dct32_fwdinv_blocks_per_wave", "adaptive_per_wave")})
print("fresh ctx defaults: %.4f ms %.3f TB/s" % t())
for a in (0, 1):
    cd.set_option("adaptive_per_wave", a)
    print("adaptive=%d: %.4f ms %.3f TB/s" % ((a,) + t()))
