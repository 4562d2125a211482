#!/usr/bin/env python3
"""Developer probe (round 5): the fused kernel's candidate launch shapes over batch sizes, next to the box's copy stream."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
NMAX = 1 << 21
x, z, y = cd.alloc(NMAX * 2048), cd.alloc(NMAX * 2048), cd.alloc(NMAX * 2048)
cd.fill_residual_dev(x.ptr, NMAX * 1024, 0x266); cd.stream_sync()
N = 20
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=8):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
cands = [(1, 0, 128, 4), (30, 16384, 256, 16), (30, 16384, 512, 16), (40, 16384, 256, 16), (20, 12288, 256, 2), (20, 12288, 128, 2), (20, 10240, 256, 2), (20, 16384, 256, 2),
         (30, 12288, 256, 3), (20, 12288, 512, 2), (20, 8192, 256, 2), (30, 16384, 64, 4), (20, 12288, 256, 1), (30, 12288, 256, 2)]
for rnd in range(2):
    for n in (1 << 20, 1000000, 1 << 21):
        t = timed(lambda: cd.mem_ceiling_dev(0, x.ptr, z.ptr, n * 2048))
        out = ["n %8d copy %.3f TB/s |" % (n, n * 4096 / t / 1e9)]
        for v, lds, tpb, bpw in cands:
            cd.set_option("dct32_fwdinv_variant", v); cd.set_option("dct32_fwdinv_lds_bytes_per_wave", lds)
            try:
                cd.set_option("dct32_wg_threads", tpb)
            except Exception:
                out.append("   -  "); continue
            cd.set_option("dct32_fwdinv_blocks_per_wave", bpw)
            t = timed(lambda: cd.dct32_fwd_inv_dev(x.ptr, z.ptr, y.ptr, n))
            out.append("%.3f" % (n * 6144 / t / 8e9))
        print(" ".join(out), flush=True)
print("columns (variant, lds, wg, blocks/wave):", cands)
