import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import x266_amd
cd = x266_amd.Codec(0)
fw = fh = 32768
nt = (fw // 16) * (fh // 16)
tc, tp = cd.alloc(nt * 512), cd.alloc(nt * 512)
cd.fill_residual_dev(tc.ptr, nt * 256, 1); cd.fill_residual_dev(tp.ptr, nt * 256, 2)
npl = fw * fh // 4
res = cd.alloc(npl * 4); cost = cd.alloc(nt * 8)
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
for r in range(3):
    print("copy %.4f  res32 %.4f  res8 %.4f  dct %.4f  satd %.4f  readprobe %.4f" % (
        timed(lambda: cd.mem_ceiling_dev(0, tc.ptr, res.ptr, nt * 256)),
        timed(lambda: cd.residual_chroma_dev(tc.ptr, tp.ptr, fw, fh, 32, res.ptr, res.ptr + npl * 2)),
        timed(lambda: cd.residual_chroma_dev(tc.ptr, tp.ptr, fw, fh, 8, res.ptr, res.ptr + npl * 2)),
        timed(lambda: cd.dct32_fwd_chroma_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, res.ptr, res.ptr + npl * 2)),
        timed(lambda: cd.satd8x8_chroma_from_tiles_dev(tc.ptr, tp.ptr, fw, fh, cost.ptr, cost.ptr + nt * 4)),
        timed(lambda: cd.mem_ceiling_dev(3, tc.ptr, cost.ptr, nt * 256))))
