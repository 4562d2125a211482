#!/usr/bin/env python3
"""Developer probe: same-box A/B (tools/ab_build.sh <git-ref>) of the STAGED SATD kernel (small and medium batches) and the two-lane frame launch of configs[4]."""
import ctypes, os, statistics
ROOT = "/root/repo"
P = ctypes.c_void_p; SZ = ctypes.c_size_t
def load(path):
    L = ctypes.CDLL(path); ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), SZ]
    L.xFillResidualDev.argtypes = [P, P, SZ, ctypes.c_uint64, ctypes.c_uint64, P]
    L.xHipStreamSync.argtypes = [P, P]
    L.xHipEventCreate.argtypes = [P, ctypes.POINTER(P)]
    L.xHipEventRecord.argtypes = [P, P, P]
    L.xHipEventElapsedMs.argtypes = [P, P, P, ctypes.POINTER(ctypes.c_double)]
    L.xSatd8x8BatchDev.argtypes = [P, P, P, SZ, P]
    L.xDct32SatdFrameDev.argtypes = [P, P, P, SZ, P, P, SZ, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [("ref", load(ROOT + "/tools/_ab/libx266hip_ref.so")), ("new", load(ROOT + "/x266_amd/libx266hip.so"))]
L0, c0, _ = libs[0][1]
def dev(n):
    p = P(); assert L0.xHipMalloc(c0, ctypes.byref(p), n) == 0; return p
x, y, d, c = dev(32400 * 2048), dev(32400 * 2048), dev(1 << 28), dev(1 << 24)
L0.xFillResidualDev(c0, x, 32400 * 1024, 1, 0, None); L0.xFillResidualDev(c0, d, 1 << 27, 2, 0, None); L0.xHipStreamSync(c0, None)
def timed(L, ctx, ev, fn, reps):
    for _ in range(20): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps * 1e3
cases = [("frame lanes 8K (32400 DCT32 + 518400 SATD)", lambda L, ctx: L.xDct32SatdFrameDev(ctx, x, y, 32400, d, c, 518400, None), 300)]
for n in (4096, 65536, 518400, 1 << 21):
    cases.append(("SATD batch %d (staged kernel)" % n, (lambda n: lambda L, ctx: L.xSatd8x8BatchDev(ctx, d, c, n, None))(n), 300))
for name, fn, reps in cases:
    r = {"ref": [], "new": []}
    for rnd in range(5):
        for tag, (L, ctx, ev) in libs: r[tag].append(timed(L, ctx, ev, fn, reps))
    print("%-46s ref %.2f us | new %.2f us | new/ref %.4f" % (name, statistics.mean(r["ref"]), statistics.mean(r["new"]), statistics.mean(r["new"]) / statistics.mean(r["ref"])), flush=True)
