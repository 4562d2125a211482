#!/usr/bin/env python3
"""Developer probe: same-box A/B of the SAD batch kernels between tools/_ab/libx266hip_ref.so (tools/ab_build.sh <git-ref>) and the working tree's library."""
import ctypes
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    L.xSadBatchDev.argtypes = [P, ctypes.c_int, P, P, P, SZ, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [("ref", load(ROOT + "/tools/_ab/libx266hip_ref.so")), ("new", load(ROOT + "/x266_amd/libx266hip.so"))]
nbytes = 1 << 30
L0, c0, _ = libs[0][1]
a, b, out = P(), P(), P()
for buf, n in ((a, nbytes), (b, nbytes), (out, nbytes // 16 * 4)): assert L0.xHipMalloc(c0, ctypes.byref(buf), n) == 0
L0.xFillResidualDev(c0, a, nbytes // 2, 1, 0, None); L0.xFillResidualDev(c0, b, nbytes // 2, 2, 0, None); L0.xHipStreamSync(c0, None)
def timed(L, ctx, ev, edge, reps=30):
    n = nbytes // (edge * edge)
    for _ in range(5): assert L.xSadBatchDev(ctx, edge, a, b, out, n, None) == 0
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): L.xSadBatchDev(ctx, edge, a, b, out, n, None)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
for tag, (L, c, ev) in libs: timed(L, c, ev, 8, 100)
for edge in (4, 8, 16, 32, 64):
    best = {"ref": [], "new": []}
    for rnd in range(5):
        for tag, (L, c, ev) in libs: best[tag].append(timed(L, c, ev, edge))
    n = nbytes // (edge * edge); byts = 2 * nbytes + 4 * n
    print("sad %2dx%-2d" % (edge, edge), " | ".join("%s mean %.4f ms frac %.3f" % (t, sum(v) / 5, byts / (sum(v) / 5) / 8e9) for t, v in best.items()), flush=True)
