#!/usr/bin/env python3
"""Developer probe: same-box A/B of the intra kernels between tools/_ab/libx266hip_ref.so (tools/ab_build.sh <git-ref>) and the working tree's
library: the predictor per mode family (all predictions of one mode) and on the bench's mix (35 modes per reference set), the 35-mode decision."""
import ctypes, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p; SZ = ctypes.c_size_t
def load(path):
    L = ctypes.CDLL(path); ctx = P()
    assert L.xHipCodecInit(ctypes.byref(ctx), 0) == 0
    L.xHipMalloc.argtypes = [P, ctypes.POINTER(P), SZ]
    L.xHipMemcpyH2D.argtypes = [P, P, P, SZ]
    L.xHipMemcpyD2H.argtypes = [P, P, P, SZ]
    L.xHipStreamSync.argtypes = [P, P]
    L.xHipEventCreate.argtypes = [P, ctypes.POINTER(P)]
    L.xHipEventRecord.argtypes = [P, P, P]
    L.xHipEventElapsedMs.argtypes = [P, P, P, ctypes.POINTER(ctypes.c_double)]
    L.xIntra32PredictDev.argtypes = [P, P, P, P, P, SZ, P]
    L.xIntra32CostsDev.argtypes = [P, P, P, P, P, SZ, P]
    ev = [P() for _ in range(2)]
    for e in ev: assert L.xHipEventCreate(ctx, ctypes.byref(e)) == 0
    return L, ctx, ev
libs = [("ref", load(ROOT + "/tools/_ab/libx266hip_ref.so")), ("new", load(ROOT + "/x266_amd/libx266hip.so"))]
L0, c0, _ = libs[0][1]
n_sets = 59918; n = n_sets * 35
rng = np.random.default_rng(5)
refs = rng.integers(0, 256, (n_sets, 144), dtype=np.uint8)
index = np.repeat(np.arange(n_sets, dtype=np.uint32), 35)
src = rng.integers(0, 256, (n_sets, 1024), dtype=np.uint8)
def dev(arr_bytes):
    p = P(); assert L0.xHipMalloc(c0, ctypes.byref(p), arr_bytes) == 0; return p
def up(p, a): assert L0.xHipMemcpyH2D(c0, p, a.ctypes.data_as(P), a.nbytes) == 0
d_refs, d_modes, d_index, d_src = dev(refs.nbytes), dev(n), dev(index.nbytes), dev(src.nbytes)
d_pred = [dev(n * 1024), dev(n * 1024)]
d_cost = [dev(n_sets * 35 * 4), dev(n_sets * 35 * 4)]; d_best = [dev(n_sets), dev(n_sets)]
up(d_refs, refs); up(d_index, index); up(d_src, src)
def timed(i, fn, reps=10):
    L, ctx, ev = libs[i][1]
    for _ in range(3): fn(L, ctx)
    ms = ctypes.c_double()
    L.xHipEventRecord(ctx, ev[0], None)
    for _ in range(reps): fn(L, ctx)
    L.xHipEventRecord(ctx, ev[1], None); L.xHipStreamSync(ctx, None)
    L.xHipEventElapsedMs(ctx, ev[0], ev[1], ctypes.byref(ms)); return ms.value / reps
def same(a, b, nbytes):
    ha, hb = np.empty(nbytes, np.uint8), np.empty(nbytes, np.uint8)
    L0.xHipMemcpyD2H(c0, ha.ctypes.data_as(P), a, nbytes); L0.xHipMemcpyD2H(c0, hb.ctypes.data_as(P), b, nbytes)
    return bool(np.array_equal(ha, hb))
cases = [("bench mix (35 modes per set)", np.tile(np.arange(35, dtype=np.uint8), n_sets))] + \
        [("all mode %2d" % m, np.full(n, m, np.uint8)) for m in (0, 1, 2, 6, 10, 14, 17, 18, 22, 26, 30, 34)]
for name, modes in cases:
    up(d_modes, modes)
    t = {"ref": [], "new": []}
    for rnd in range(3):
        for i, (tag, _) in enumerate(libs):
            t[tag].append(timed(i, lambda L, ctx: L.xIntra32PredictDev(ctx, d_refs, d_modes, d_index, d_pred[i], n, None)))
    L0.xHipStreamSync(c0, None)
    r, w = min(t["ref"]), min(t["new"])
    print("predict %-30s ref %.4f ms (%.2f TB/s)  new %.4f ms (%.2f TB/s)  new/ref %.3f  identical %s" % (name, r, n * 1024 / r / 1e9, w, n * 1024 / w / 1e9, w / r, same(d_pred[0], d_pred[1], n * 1024)), flush=True)
t = {"ref": [], "new": []}
for rnd in range(3):
    for i, (tag, _) in enumerate(libs):
        t[tag].append(timed(i, lambda L, ctx: L.xIntra32CostsDev(ctx, d_refs, d_src, d_cost[i], d_best[i], n_sets, None)))
r, w = min(t["ref"]), min(t["new"])
print("decide 35 modes, %d blocks: ref %.4f ms  new %.4f ms  new/ref %.3f  identical costs %s modes %s" % (n_sets, r, w, w / r, same(d_cost[0], d_cost[1], n_sets * 140), same(d_best[0], d_best[1], n_sets)))
