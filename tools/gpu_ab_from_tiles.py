#!/usr/bin/env python3
"""Developer probe: same-box A/B of xSatd8x8FromTilesDev between tools/_ab/libx266hip_ref.so (tools/ab_build.sh <git-ref>) and the working tree's library."""
import ctypes, os, sys
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
    L.xSatd8x8FromTilesDev.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [("ref", load(ROOT + "/tools/_ab/libx266hip_ref.so")), ("new", load(ROOT + "/x266_amd/libx266hip.so"))]
w = h = 32768
nt = (w // 16) * (h // 16)
L0, c0, _ = libs[0][1]
cur, pred, out = P(), P(), P()
for b, n in ((cur, nt * 512), (pred, nt * 512), (out, w * h // 64 * 4)): assert L0.xHipMalloc(c0, ctypes.byref(b), n) == 0
L0.xFillResidualDev(c0, cur, nt * 256, 1, 0, None); L0.xFillResidualDev(c0, pred, nt * 256, 2, 0, None); L0.xHipStreamSync(c0, None)
def timed(L, ctx, ev, reps=30):
    for _ in range(5): L.xSatd8x8FromTilesDev(ctx, cur, pred, w, h, out, None)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): L.xSatd8x8FromTilesDev(ctx, cur, pred, w, h, out, None)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
for tag, (L, c, ev) in libs: timed(L, c, ev, 100)
best = {"ref": [], "new": []}
for rnd in range(6):
    for tag, (L, c, ev) in libs: best[tag].append(timed(L, c, ev))
for tag in best: print(tag, "min %.4f mean %.4f ms" % (min(best[tag]), sum(best[tag]) / 6), "frac at mean %.3f" % (w * h // 64 * 132 / (sum(best[tag]) / 6) / 8e9))
