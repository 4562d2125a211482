import ctypes, os, sys, statistics
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import x266_amd
libs = [("shipped", None), ("no position transforms", os.path.join(R, "tools/_ab/libx266hip_ablate1.so")), ("no ds_min in the loop (keys kept alive)", os.path.join(R, "tools/_ab/libx266hip_ablate2.so")), ("no >> 2", os.path.join(R, "tools/_ab/libx266hip_ablate3.so"))]
w, h, rng = 3840, 2160, 64
rs = np.random.RandomState(3)
cur = rs.randint(0, 256, (h, w)).astype(np.uint8); refp = rs.randint(0, 256, (h + 2 * rng, w + 2 * rng)).astype(np.uint8)
res = {}
cds = [(n, x266_amd.Codec(0, library=p)) for n, p in libs]
bufs = []
for n, cd in cds:
    dc, dr = cd.alloc(cur.nbytes), cd.alloc(refp.nbytes); dc.upload(cur); dr.upload(refp)
    best = cd.alloc((w // 8) * (h // 8) * 8)
    bufs.append((dc, dr, best))
for rnd in range(3):
    row = []
    for (n, cd), (dc, dr, best) in zip(cds, bufs):
        ev = [cd.event_create() for _ in range(11)]
        fn = lambda: cd.satd_search_dev(dc.ptr, w, dr.ptr + rng * refp.shape[1] + rng, refp.shape[1], w, h, rng, best.ptr)
        for _ in range(5): fn()
        cd.stream_sync()
        for i in range(10):
            cd.event_record(ev[i]); fn()
        cd.event_record(ev[10]); cd.stream_sync()
        row.append("%s %.4f ms" % (n, statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(10)])))
    print(" | ".join(row))
