#!/usr/bin/env python3
"""Developer probe: same-box, same-process A/B of two builds of libx266hip.so (GPU box).
A = tools/_ab/libx266hip_ref.so (tools/ab_build.sh <git-ref>), B = the working tree's library.
Alternates the two so that clock state and box are shared."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p
def load(path):
    L = ctypes.CDLL(path)
    ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), ctypes.c_size_t]
    L.xFillResidualDev.argtypes = [P, P, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64, P]
    L.xHipTimeKernel.argtypes = [P, ctypes.c_int, P, P, ctypes.c_size_t, ctypes.c_int, P, ctypes.POINTER(ctypes.c_double)]
    L.xHipStreamSync.argtypes = [P, P]
    return L, ctx
libs = [("ref", load(os.path.join(ROOT, "tools", "_ab", "libx266hip_ref.so"))), ("new", load(os.path.join(ROOT, "x266_amd", "libx266hip.so")))]
N = 1 << 20
L0, c0 = libs[0][1]
din, dout = P(), P()
assert L0.xHipMalloc(c0, ctypes.byref(din), N * 2048) == 0 and L0.xHipMalloc(c0, ctypes.byref(dout), N * 2048) == 0
L0.xFillResidualDev(c0, din, N * 1024, 0x266, 0, None); L0.xHipStreamSync(c0, None)
def t(L, ctx, op, n, reps=20):
    ms = ctypes.c_double()
    assert L.xHipTimeKernel(ctx, op, din, dout, n, reps, None, ctypes.byref(ms)) == 0
    return ms.value
dre = P()
assert L0.xHipMalloc(c0, ctypes.byref(dre), N * 2048) == 0
def t_fused(L, ctx, reps=20):
    L.xDct32FwdInvBatchDev.argtypes = [P, P, P, P, ctypes.c_size_t, P]
    for _ in range(3): L.xDct32FwdInvBatchDev(ctx, din, dout, dre, N, None)
    L.xHipStreamSync(ctx, None)
    t0 = time.perf_counter()
    for _ in range(reps): L.xDct32FwdInvBatchDev(ctx, din, dout, dre, N, None)
    L.xHipStreamSync(ctx, None)
    return (time.perf_counter() - t0) / reps * 1e3
t(L0, c0, 0, N, 150)                                      # warm the clocks
for op, name, n, unit in ((0, "fwd", N, 4096), (1, "inv", N, 4096), (2, "satd", 1 << 24, 132)):
    best = {"ref": 1e9, "new": 1e9}
    for rnd in range(6):
        for tag, (L, ctx) in libs:
            best[tag] = min(best[tag], t(L, ctx, op, n))
    print("%-4s ref %.4f ms %.3f TB/s | new %.4f ms %.3f TB/s | new/ref time %.4f" % (
        name, best["ref"], n * unit / best["ref"] / 1e9, best["new"], n * unit / best["new"] / 1e9, best["new"] / best["ref"]), flush=True)
best = {"ref": 1e9, "new": 1e9}
for rnd in range(6):
    for tag, (L, ctx) in libs:
        best[tag] = min(best[tag], t_fused(L, ctx))
print("fused ref %.4f ms %.3f TB/s | new %.4f ms %.3f TB/s | new/ref time %.4f" % (
    best["ref"], N * 6144 / best["ref"] / 1e9, best["new"], N * 6144 / best["new"] / 1e9, best["new"] / best["ref"]), flush=True)

# ---- motion search (one 4K frame, +-64) and intra mode decision (2^17 blocks), wall-clock
w, h, rng = 3840, 2160, 64
stride = w + 2 * rng + 16
cur, refp, bestb = P(), P(), P()
L0.xHipMalloc(c0, ctypes.byref(cur), w * h); L0.xHipMalloc(c0, ctypes.byref(refp), stride * (h + 2 * rng + 16)); L0.xHipMalloc(c0, ctypes.byref(bestb), (w // 8) * (h // 8) * 8)
L0.xFillResidualDev(c0, cur, w * h // 2, 1, 0, None); L0.xFillResidualDev(c0, refp, stride * (h + 2 * rng + 16) // 2, 2, 0, None)
nb = 1 << 17
irefs, isrc, icost, ibest = P(), P(), P(), P()
L0.xHipMalloc(c0, ctypes.byref(irefs), nb * 144); L0.xHipMalloc(c0, ctypes.byref(isrc), nb * 1024); L0.xHipMalloc(c0, ctypes.byref(icost), nb * 35 * 4); L0.xHipMalloc(c0, ctypes.byref(ibest), nb)
L0.xFillResidualDev(c0, irefs, nb * 72, 3, 0, None); L0.xFillResidualDev(c0, isrc, nb * 512, 4, 0, None); L0.xHipStreamSync(c0, None)
def wall(fn, L, ctx, reps):
    for _ in range(2): fn(L, ctx)
    L.xHipStreamSync(ctx, None); t0 = time.perf_counter()
    for _ in range(reps): fn(L, ctx)
    L.xHipStreamSync(ctx, None); return (time.perf_counter() - t0) / reps * 1e3
def me(L, ctx):
    L.xSatd8x8SearchDev.argtypes = [P, P, ctypes.c_ssize_t, P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P]
    assert L.xSatd8x8SearchDev(ctx, cur, w, P(refp.value + rng * stride + rng), stride, w, h, rng, bestb, None, None) == 0
def intra(L, ctx):
    L.xIntra32CostsDev.argtypes = [P, P, P, P, P, ctypes.c_size_t, P]
    assert L.xIntra32CostsDev(ctx, irefs, isrc, icost, ibest, nb, None) == 0
def sadme(L, ctx):
    L.xSad8x8SearchDev.argtypes = [P, P, ctypes.c_ssize_t, P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P]
    assert L.xSad8x8SearchDev(ctx, cur, w, P(refp.value + rng * stride + rng), stride, w, h, rng, bestb, None, None) == 0
for name, fn, reps in (("me 4K", me, 5), ("intra decide", intra, 10), ("sad me 4K", sadme, 5)):
    best = {"ref": 1e9, "new": 1e9}
    for rnd in range(4):
        for tag, (L, ctx) in libs:
            if (hasattr(L, "xIntra32CostsDev") or name != "intra decide") and (hasattr(L, "xSad8x8SearchDev") or name != "sad me 4K"):
                best[tag] = min(best[tag], wall(fn, L, ctx, reps))
    print("%-12s ref %.4f ms | new %.4f ms | new/ref time %.4f" % (name, best["ref"], best["new"], best["new"] / best["ref"]), flush=True)
