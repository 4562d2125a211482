#!/usr/bin/env python3
"""Developer probe: same-box, same-process A/B of the inverse-transform family between tools/_ab/libx266hip_ref.so
(tools/ab_build.sh <git-ref>) and the working tree's library: DCT32 inverse, fused forward+inverse, the small-N inverses,
the one-launch tile inverse.  HIP events per launch (xHipEvent*), best of alternating rounds."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p
SZ = ctypes.c_size_t
def load(path):
    L = ctypes.CDLL(path)
    ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), SZ]
    L.xFillResidualDev.argtypes = [P, P, SZ, ctypes.c_uint64, ctypes.c_uint64, P]
    L.xHipStreamSync.argtypes = [P, P]
    L.xHipEventCreate.argtypes = [P, ctypes.POINTER(P)]
    L.xHipEventRecord.argtypes = [P, P, P]
    L.xHipEventElapsedMs.argtypes = [P, P, P, ctypes.POINTER(ctypes.c_double)]
    L.xDct32InvBatchDev.argtypes = [P, P, P, SZ, P]
    L.xDct32FwdInvBatchDev.argtypes = [P, P, P, P, SZ, P]
    L.xTransformInvBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformTilesDev.argtypes = [P, ctypes.c_int, P, P, SZ, P, P, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [("ref", load(os.path.join(ROOT, "tools", "_ab", "libx266hip_ref.so"))), ("new", load(os.path.join(ROOT, "x266_amd", "libx266hip.so")))]
N = 1 << 20
L0, c0, _ = libs[0][1]
din, dout, dre, dcls = P(), P(), P(), P()
for b, n in ((din, N * 2048), (dout, N * 2048), (dre, N * 2048), (dcls, N)):
    assert L0.xHipMalloc(c0, ctypes.byref(b), n) == 0
L0.xFillResidualDev(c0, din, N * 1024, 0x266, 0, None)
import numpy as np
cls = np.array([3, 2, 6, 1, 5, 0, 4], np.uint8)[(np.arange(N) + np.arange(N) // 4) % 7]
L0.xHipMemcpyH2D.argtypes = [P, P, P, SZ]
assert L0.xHipMemcpyH2D(c0, dcls, cls.ctypes.data_as(P), N) == 0
L0.xHipStreamSync(c0, None)
def timed(L, ctx, ev, fn, reps=20):
    for _ in range(3): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None)
    L.xHipStreamSync(ctx, None)
    assert L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)) == 0
    return ms.value / reps
legs = [("dct32 inverse", 4096, lambda L, c: L.xDct32InvBatchDev(c, din, dout, N, None)),
        ("fused fwd+inv", 6144, lambda L, c: L.xDct32FwdInvBatchDev(c, din, dout, dre, N, None))]
for n in (4, 8, 16):
    per = (32 // n) ** 2
    legs.append(("dct2 inverse %dx%d" % (n, n), 4096, (lambda n, per: lambda L, c: L.xTransformInvBatchDev(c, 0, n, din, dout, N * per, None, None))(n, per)))
legs.append(("tiles one launch fwd", 4096, lambda L, c: L.xTransformTilesDev(c, 0, din, dout, N, None, dcls, None)))
legs.append(("tiles one launch inv", 4096, lambda L, c: L.xTransformTilesDev(c, 1, din, dout, N, None, dcls, None)))
for tag, (L, c, ev) in libs: timed(L, c, ev, legs[0][2], 100)     # clocks
for name, unit, fn in legs:
    best = {"ref": 1e9, "new": 1e9}
    for rnd in range(6):
        for tag, (L, c, ev) in libs:
            best[tag] = min(best[tag], timed(L, c, ev, fn))
    print("%-22s ref %.4f ms frac %.3f | new %.4f ms frac %.3f | new/ref time %.4f" % (
        name, best["ref"], N * unit / best["ref"] / 8e9, best["new"], N * unit / best["new"] / 8e9, best["new"] / best["ref"]), flush=True)
