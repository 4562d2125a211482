#!/usr/bin/env python3
"""Developer probe (round 5): the offset-table paths of the transform set (xTransformFwd/InvBatchDev with d_offsets, xTransformTilesDev with tile offsets) ship with plain
loads and stores; four builds (tools/ab_build_policies.sh indexed: 0 shipped, 1 nt loads + 'sc1 nt' stores, 2 stores only, 3 loads only) in one process on the same buffers.
Layout: a CTU-ordered buffer of 2^20 regions of 1024 samples whose classes cycle through seven (type, size) classes; a class's call gets the offsets of ITS regions' blocks."""
import ctypes, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = ctypes.c_void_p; SZ = ctypes.c_size_t
NAMES = ["shipped (plain)", "nt loads + sc1 nt stores", "sc1 nt stores", "nt loads"]
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
    L.xTransformFwdBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformInvBatchDev.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, SZ, P, P]
    L.xTransformTilesDev.argtypes = [P, ctypes.c_int, P, P, SZ, P, P, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [load("%s/tools/_ab/libx266hip_ix%d.so" % (ROOT, k)) for k in range(4)]
L0, c0, _ = libs[0]
def dev(nb):
    p = P(); assert L0.xHipMalloc(c0, ctypes.byref(p), nb) == 0; return p
def up(a):
    p = dev(a.nbytes); assert L0.xHipMemcpyH2D(c0, p, a.ctypes.data_as(P), a.nbytes) == 0; return p
def timed(lib, fn, reps=10):
    L, ctx, ev = lib
    for _ in range(3): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
n = 1 << 20
CLASSES = [(0, 4), (0, 8), (0, 16), (0, 32), (1, 4), (1, 8), (1, 16)]
q = np.arange(n)
cls_of = (q + q // 4) % 7
x, z = dev(n * 2048), dev(n * 2048)
L0.xFillResidualDev(c0, x, n * 1024, 0x266, 0, None); L0.xHipStreamSync(c0, None)
calls = []
for k, (ttype, size) in enumerate(CLASSES):
    regions = q[cls_of == k].astype(np.uint64) * 1024
    per = 1024 // (size * size)
    offs = (regions[:, None] + np.arange(per, dtype=np.uint64)[None, :] * (size * size)).ravel().astype(np.uint32)
    calls.append((ttype, size, up(offs), offs.size))
tile_class = up(np.array([t * 4 + {4: 0, 8: 1, 16: 2, 32: 3}[s] for t, s in CLASSES], np.uint8)[cls_of])
perm = np.random.default_rng(1).permutation(n).astype(np.uint32)
tile_offs_sorted, tile_offs_perm = up((q * 1024).astype(np.uint32)), up(perm * 1024)
def seven(inv):
    def f(L, c):
        for ttype, size, offs, cnt in calls:
            (L.xTransformInvBatchDev if inv else L.xTransformFwdBatchDev)(c, ttype, size, x, z, cnt, offs, None)
    return f
cases = [("seven per-class calls over offset tables, forward", seven(0)), ("... inverse", seven(1)),
         ("tile kernel, identity offset table, forward", lambda L, c: L.xTransformTilesDev(c, 0, x, z, n, tile_offs_sorted, tile_class, None)),
         ("tile kernel, permuted offset table, forward", lambda L, c: L.xTransformTilesDev(c, 0, x, z, n, tile_offs_perm, tile_class, None)),
         ("tile kernel, permuted offset table, inverse", lambda L, c: L.xTransformTilesDev(c, 1, x, z, n, tile_offs_perm, tile_class, None)),
         ("tile kernel, no offset table (contiguous), forward", lambda L, c: L.xTransformTilesDev(c, 0, x, z, n, None, tile_class, None))]
for name, fn in cases:
    r = [[] for _ in libs]
    for rnd in range(3):
        for k, lib in enumerate(libs): r[k].append(timed(lib, fn))
    m = [statistics.median(v) for v in r]
    print("%-52s " % name + "  ".join("%s %.4f (%+.1f%%)" % (NAMES[k], m[k], 100 * (m[k] / m[0] - 1)) for k in range(4)), flush=True)
