#!/usr/bin/env python3
"""Developer probe (round 5): the cache-policy bits of the STREAM stores (store16_sc1nt / store_tile_sc1nt: every transform, intra and tile kernel's outputs) -- six builds
of the library (tools/_ab/libx266hip_pol<k>.so: 0 'sc1 nt' [shipped], 1 'sc1', 2 'nt', 3 plain, 4 'sc0 sc1', 5 'sc0 sc1 nt') timed in ONE process on the same buffers,
alternating rounds, several allocation sets."""
import ctypes, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = ctypes.c_void_p; SZ = ctypes.c_size_t
NAMES = ["sc1 nt", "sc1", "nt", "plain", "sc0 sc1", "sc0 sc1 nt"]
def load(path):
    L = ctypes.CDLL(path); ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), SZ]
    L.xHipMemcpyH2D.argtypes = [P, P, P, SZ]
    L.xFillResidualDev.argtypes = [P, P, SZ, ctypes.c_uint64, ctypes.c_uint64, P]
    L.xHipStreamSync.argtypes = [P, P]
    L.xHipEventCreate.argtypes = [P, ctypes.POINTER(P)]
    L.xHipEventRecord.argtypes = [P, P, P]
    L.xHipEventElapsedMs.argtypes = [P, P, P, ctypes.POINTER(ctypes.c_double)]
    L.xDct32FwdBatchDev.argtypes = [P, P, P, SZ, P]
    L.xDct32InvBatchDev.argtypes = [P, P, P, SZ, P]
    L.xDct32FwdInvBatchDev.argtypes = [P, P, P, P, SZ, P]
    L.xTransformFwdBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformInvBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xIntra32PredictDev.argtypes = [P, P, P, P, P, SZ, P]
    L.xIntra32ResidualDct32Dev.argtypes = [P, P, P, P, P, P, SZ, P]
    L.xHipMemCeilingDev.argtypes = [P, ctypes.c_int, P, P, SZ, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [load("%s/tools/_ab/libx266hip_pol%d.so" % (ROOT, k)) for k in range(6)]
L0, c0, _ = libs[0]
def dev(nb):
    p = P(); assert L0.xHipMalloc(c0, ctypes.byref(p), nb) == 0; return p
n = 1 << 20
n_sets = (n + 34) // 35
refs = dev(n_sets * 144); L0.xFillResidualDev(c0, refs, n_sets * 72, 3, 0, None)
modes_h = np.tile(np.arange(35, dtype=np.uint8), n_sets)[:n].copy(); index_h = np.repeat(np.arange(n_sets, dtype=np.uint32), 35)[:n].copy()
modes, index = dev(n), dev(4 * n)
L0.xHipMemcpyH2D(c0, modes, modes_h.ctypes.data_as(P), n); L0.xHipMemcpyH2D(c0, index, index_h.ctypes.data_as(P), 4 * n)
def timed(lib, fn, reps=12):
    L, ctx, ev = lib
    for _ in range(4): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
keep = []
for aset in range(3):
    x, z, y = dev(n * 2048), dev(n * 2048), dev(n * 2048); keep += [x, z, y]
    L0.xFillResidualDev(c0, x, n * 1024, 0x266, 0, None); L0.xHipStreamSync(c0, None)
    cases = [("copy stream", lambda L, c: L.xHipMemCeilingDev(c, 0, x, z, n * 2048, None)), ("write stream", lambda L, c: L.xHipMemCeilingDev(c, 2, x, z, n * 2048, None)),
             ("forward DCT32", lambda L, c: L.xDct32FwdBatchDev(c, x, z, n, None)), ("inverse DCT32", lambda L, c: L.xDct32InvBatchDev(c, x, z, n, None)),
             ("fused fwd+inv", lambda L, c: L.xDct32FwdInvBatchDev(c, x, z, y, n, None)), ("reconstruction only", lambda L, c: L.xDct32FwdInvBatchDev(c, x, None, y, n, None)),
             ("DCT-II 8x8 forward", lambda L, c: L.xTransformFwdBatchDev(c, 0, 8, x, z, n * 16, None, None)), ("DST-VII 4x4 inverse", lambda L, c: L.xTransformInvBatchDev(c, 1, 4, x, z, n * 64, None, None)),
             ("intra predict (35-mode mix)", lambda L, c: L.xIntra32PredictDev(c, refs, modes, index, z, n, None)),
             ("intra predict->residual->DCT32", lambda L, c: L.xIntra32ResidualDct32Dev(c, refs, modes, index, x, z, n, None))]
    for name, fn in cases:
        r = [[] for _ in libs]
        for rnd in range(3):
            for k, lib in enumerate(libs): r[k].append(timed(lib, fn))
        m = [statistics.median(v) for v in r]
        print("set %d %-32s " % (aset, name) + "  ".join("%s %.4f (%+.1f%%)" % (NAMES[k], m[k], 100 * (m[k] / m[0] - 1)) for k in range(6)), flush=True)
    keep.append(dev((aset + 1) * 411 << 20))
