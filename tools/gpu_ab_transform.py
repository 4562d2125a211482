#!/usr/bin/env python3
"""Developer probe: same-box A/B of the transform-set kernels (per-class forward / inverse, the one-launch mixed tile call) between
tools/_ab/libx266hip_ref.so (tools/ab_build.sh <git-ref>) and the working tree's library; alternating rounds, best and median of each."""
import ctypes, os, statistics, sys, time
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
    L.xHipMemcpyH2D.argtypes = [P, P, P, SZ]
    L.xTransformFwdBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformInvBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformTilesDev.argtypes = [P, ctypes.c_int, P, P, SZ, P, P, P]
    L.xDct32FwdBatchDev.argtypes = [P, P, P, SZ, P]
    return L, ctx


libs = [("ref", load(os.path.join(ROOT, "tools", "_ab", "libx266hip_ref.so"))), ("new", load(os.path.join(ROOT, "x266_amd", "libx266hip.so")))]
N = 1 << 20
L0, c0 = libs[0][1]
din, dout, dcls = P(), P(), P()
assert L0.xHipMalloc(c0, ctypes.byref(din), N * 2048) == 0 and L0.xHipMalloc(c0, ctypes.byref(dout), N * 2048) == 0 and L0.xHipMalloc(c0, ctypes.byref(dcls), N) == 0
L0.xFillResidualDev(c0, din, N * 1024, 0x266, 0, None)
cls = bytes([[3, 2, 6, 1, 5, 0, 4][(q + q // 4) % 7] for q in range(N)])
L0.xHipMemcpyH2D(c0, dcls, cls, N)
L0.xHipStreamSync(c0, None)


def wall(L, ctx, fn, reps=20):
    for _ in range(3):
        fn()
    L.xHipStreamSync(ctx, None)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    L.xHipStreamSync(ctx, None)
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(200):
    L0.xDct32FwdBatchDev(c0, din, dout, N, None)
L0.xHipStreamSync(c0, None)
cases = [("dct32 fwd", lambda L, c: L.xDct32FwdBatchDev(c, din, dout, N, None))]
for n in (4, 8, 16):
    cnt = N * 1024 // (n * n)
    cases.append(("dct2 %dx%d fwd" % (n, n), lambda L, c, n=n, cnt=cnt: L.xTransformFwdBatchDev(c, 0, n, din, dout, cnt, None, None)))
    cases.append(("dct2 %dx%d inv" % (n, n), lambda L, c, n=n, cnt=cnt: L.xTransformInvBatchDev(c, 0, n, din, dout, cnt, None, None)))
cases.append(("tiles fwd", lambda L, c: L.xTransformTilesDev(c, 0, din, dout, N, None, dcls, None)))
cases.append(("tiles inv", lambda L, c: L.xTransformTilesDev(c, 1, din, dout, N, None, dcls, None)))
for name, fn in cases:
    t = {"ref": [], "new": []}
    for rnd in range(7):
        for tag, (L, ctx) in libs:
            t[tag].append(wall(L, ctx, lambda: fn(L, ctx)))
    print("%-16s ref best %.4f median %.4f ms | new best %.4f median %.4f ms | new/ref median %.4f   (%.3f -> %.3f of 8 TB/s)" % (
        name, min(t["ref"]), statistics.median(t["ref"]), min(t["new"]), statistics.median(t["new"]), statistics.median(t["new"]) / statistics.median(t["ref"]),
        N * 4096 / statistics.median(t["ref"]) / 8e9, N * 4096 / statistics.median(t["new"]) / 8e9), flush=True)
