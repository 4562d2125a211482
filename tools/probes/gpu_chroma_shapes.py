#!/usr/bin/env python3
"""Developer probe (round 6): the chroma kernels over launch shapes, PAIRED in one process on the same buffers
(a probe build of the launchers read X266_CH_* / X266_CD_* / X266_CS_* on every launch: workgroup threads, LDS charge, kernel variant;
the shipped launchers carry the winners as constants -- result in profiles/r06_chroma_shapes.txt; gpu_chroma_time.py times the shipped shapes)."""
import os, sys, statistics, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import x266_amd
cd = x266_amd.Codec(0)
fw = fh = 32768
nt = (fw // 16) * (fh // 16)
tc, tp = cd.alloc(nt * 512), cd.alloc(nt * 512)
cd.fill_residual_dev(tc.ptr, nt * 256, 1); cd.fill_residual_dev(tp.ptr, nt * 256, 2)
npl = fw * fh // 4
res = cd.alloc(npl * 4)
cost = cd.alloc(nt * 8)
cd.stream_sync()
N = 14
ev = [cd.event_create() for _ in range(N + 1)]
def timed(fn, warm=6):
    for _ in range(warm): fn()
    cd.stream_sync()
    for i in range(N):
        cd.event_record(ev[i]); fn()
    cd.event_record(ev[N]); cd.stream_sync()
    return statistics.median([cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)])
def env(**kw):
    for k, v in kw.items(): os.environ[k] = str(v)
GB = nt * 256 * 2 / 1e9        # bytes moved by every chroma kernel on this frame except the SATD (reads only: nt*256 + costs)
for rnd in range(2):
    cp = timed(lambda: cd.mem_ceiling_dev(0, tc.ptr, res.ptr, nt * 256))
    print("round %d: copy of the same bytes %.4f ms (%.3f TB/s)" % (rnd, cp, GB / cp))
    rows = []
    for v, wg, lds in itertools.product((0, 1), (64, 128, 256), (8192, 12288, 16384, 20480, 32768)):
        if lds * 1 > 65536: continue
        env(X266_CH_V=v, X266_CH_WG=wg, X266_CH_LDS=lds)
        a = timed(lambda: cd.residual_chroma_dev(tc.ptr, tp.ptr, fw, fh, 32, res.ptr, res.ptr + npl * 2))
        b = timed(lambda: cd.residual_chroma_dev(tc.ptr, tp.ptr, fw, fh, 8, res.ptr, res.ptr + npl * 2)) if v == 0 else 0
        rows.append((a, b, v, wg, lds))
    rows.sort()
    print("  residual 32: " + "  ".join("%.4f (v%d wg%d lds%d)" % (r[0], r[2], r[3], r[4]) for r in rows[:6]))
    r8 = sorted(r for r in rows if r[2] == 0)
    r8.sort(key=lambda r: r[1])
    print("  residual 8 : " + "  ".join("%.4f (wg%d lds%d)" % (r[1], r[3], r[4]) for r in r8[:6]))
    rows = []
    for v, wg, lds in itertools.product((0, 1), (64, 128, 256), (4096, 6144, 8192, 10240, 12288, 16384)):
        if v == 1 and lds < 4096: continue
        env(X266_CD_V=v, X266_CD_WG=wg, X266_CD_LDS=lds)
        rows.append((timed(lambda: cd.dct32_fwd_chroma_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, res.ptr, res.ptr + npl * 2)), v, wg, lds))
    rows.sort()
    print("  chroma dct : " + "  ".join("%.4f (v%d wg%d lds%d)" % r for r in rows[:8]))
    rows = []
    for wg, lds in itertools.product((64, 128, 256), (4096, 6144, 8192, 10240, 12288, 16384)):
        env(X266_CS_WG=wg, X266_CS_LDS=lds)
        rows.append((timed(lambda: cd.satd8x8_chroma_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, cost.ptr, cost.ptr + nt * 4)), wg, lds))
    rows.sort()
    print("  chroma satd: " + "  ".join("%.4f (wg%d lds%d)" % r for r in rows[:8]))
    rd = timed(lambda: cd.mem_ceiling_dev(3, tc.ptr, cost.ptr, nt * 256))
    print("  read probe of the same bytes (dense) %.4f ms" % rd)
