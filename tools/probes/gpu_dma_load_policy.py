#!/usr/bin/env python3
"""Developer probe (round 5): the cache-policy bits of the LDS-DMA loads (global_load_lds_dwordx4 ... nt: SATD batch, fused DCT32, from-tiles SATD) -- six builds of the
library (tools/ab_build_policies.sh dma-load: 0 'nt' [shipped], 1 plain, 2 'sc1', 3 'sc0 sc1', 4 'sc1 nt', 5 'sc0') timed in ONE process on the same buffers."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = ctypes.c_void_p; SZ = ctypes.c_size_t
NAMES = ["nt", "plain", "sc1", "sc0 sc1", "sc1 nt", "sc0"]
def load(path):
    L = ctypes.CDLL(path); ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), SZ]
    L.xFillResidualDev.argtypes = [P, P, SZ, ctypes.c_uint64, ctypes.c_uint64, P]
    L.xHipStreamSync.argtypes = [P, P]
    L.xHipEventCreate.argtypes = [P, ctypes.POINTER(P)]
    L.xHipEventRecord.argtypes = [P, P, P]
    L.xHipEventElapsedMs.argtypes = [P, P, P, ctypes.POINTER(ctypes.c_double)]
    L.xDct32FwdInvBatchDev.argtypes = [P, P, P, P, SZ, P]
    L.xSatd8x8BatchDev.argtypes = [P, P, P, SZ, P]
    L.xSatd8x8FromTilesDev.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, P, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [load("%s/tools/_ab/libx266hip_ld%d.so" % (ROOT, k)) for k in range(6)]
L0, c0, _ = libs[0]
def dev(nb):
    p = P(); assert L0.xHipMalloc(c0, ctypes.byref(p), nb) == 0; return p
def timed(lib, fn, reps=12):
    L, ctx, ev = lib
    for _ in range(4): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
n = 1 << 20
keep = []
for aset in range(3):
    x, z, y, c = dev(n * 2048), dev(n * 2048), dev(n * 2048), dev(1 << 26); keep += [x, z, y, c]
    L0.xFillResidualDev(c0, x, n * 1024, 0x266, 0, None); L0.xFillResidualDev(c0, z, n * 1024, 0x267, 0, None); L0.xHipStreamSync(c0, None)
    cases = [("SATD batch 2^24", lambda L, cx: L.xSatd8x8BatchDev(cx, x, c, 1 << 24, None)), ("fused fwd+inv", lambda L, cx: L.xDct32FwdInvBatchDev(cx, x, z, y, n, None)),
             ("reconstruction only", lambda L, cx: L.xDct32FwdInvBatchDev(cx, x, None, y, n, None)),
             ("from-tiles SATD 32768^2", lambda L, cx: L.xSatd8x8FromTilesDev(cx, x, z, 32768, 32768, c, None))]
    for name, fn in cases:
        r = [[] for _ in libs]
        for rnd in range(3):
            for k, lib in enumerate(libs): r[k].append(timed(lib, fn))
        m = [statistics.median(v) for v in r]
        print("set %d %-24s " % (aset, name) + "  ".join("%s %.4f (%+.1f%%)" % (NAMES[k], m[k], 100 * (m[k] / m[0] - 1)) for k in range(6)), flush=True)
    L0.xFillResidualDev(c0, z, n * 1024, 0x267, 0, None)
    keep.append(dev((aset + 1) * 411 << 20))
