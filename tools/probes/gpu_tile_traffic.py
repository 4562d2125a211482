#!/usr/bin/env python3
"""Developer probe (round 6): HBM traffic of the tile-stage kernels from the PMC counters -- do the half-dense (luma: 256 of every 512 bytes) and
quarter-dense (chroma: one 128-byte line of every four) reads fetch only the lines they use?
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d out_f -- python tools/probes/gpu_tile_traffic.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d out_w -- python tools/probes/gpu_tile_traffic.py run
    python tools/probes/gpu_tile_traffic.py report out_f out_w
The copy stream of known volume in the same passes calibrates the counters' units (gfx950 counts a 128-byte request of a 16 B-per-lane load as 64 B)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
W = H = 16384
NT = (W // 16) * (H // 16)


def run():
    import x266_amd
    cd = x266_amd.Codec(0)
    tc, tp = cd.alloc(NT * 512), cd.alloc(NT * 512)
    cd.fill_residual_dev(tc.ptr, NT * 256, 1); cd.fill_residual_dev(tp.ptr, NT * 256, 2)
    out = cd.alloc(W * H * 3)
    for _ in range(3):
        cd.mem_ceiling_dev(0, tc.ptr, out.ptr, NT * 512)                                   # calibration: reads and writes exactly NT * 512 bytes
        cd.residual_luma_dev(tc.ptr, tp.ptr, W, H, 32, out.ptr)
        cd.dct32_fwd_from_tiles_dev(tc.ptr, tp.ptr, W, H, out.ptr)
        cd.satd8x8_from_tiles_dev(tc.ptr, tp.ptr, W, H, out.ptr)
        cd.residual_chroma_dev(tc.ptr, tp.ptr, W, H, 32, out.ptr, out.ptr + W * H // 2)
        cd.residual_chroma_dev(tc.ptr, tp.ptr, W, H, 8, out.ptr, out.ptr + W * H // 2)
        cd.dct32_fwd_chroma_from_tiles_dev(tc.ptr, tp.ptr, W, H, out.ptr, out.ptr + W * H // 2)
        cd.satd8x8_chroma_from_tiles_dev(tc.ptr, tp.ptr, W, H, out.ptr, out.ptr + NT * 4)
        cd.dct32_fwd_ctu_from_tiles_dev(tc.ptr, tp.ptr, W, H, out.ptr)
    cd.stream_sync()


def mean_by_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "x266" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]        # the kernel's own name with its template arguments
                acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def report(df, dw):
    f, w = mean_by_kernel(df, "FETCH_SIZE"), mean_by_kernel(dw, "WRITE_SIZE")
    cal = next(k for k in f if k.startswith("mem_ceiling_kernel<0"))
    kf, kw = NT * 512 / f[cal], NT * 512 / w[cal]                                            # bytes per counter unit, from the copy of known volume
    px = W * H
    alg = {"residual_luma_kernel<5": (px * 2, px * 2), "dct32_from_tiles_kernel": (px * 2, px * 2), "satd8x8_from_tiles": (px * 2, px // 16),
           "residual_chroma_kernel<5": (px, px), "residual_chroma_kernel<3": (px, px), "dct32_chroma_from_tiles_kernel": (px, px),
           "satd8x8_chroma_from_tiles_kernel": (px, NT * 8), "dct32_ctu_from_tiles_kernel": (px * 3, px * 3)}
    print("# 16384 x 16384 tiled frame pair; bytes read / written per launch from FETCH_SIZE / WRITE_SIZE (units calibrated on the copy: %.1f / %.1f B) over the algorithmic bytes" % (kf, kw))
    for k in sorted(f):
        a = next((v for p, v in alg.items() if k.startswith(p)), None)
        if a:
            print("%-44s read %.4f GB = %.3f x algorithmic   written %.4f GB = %.3f x algorithmic" % (k, f[k] * kf / 1e9, f[k] * kf / a[0], w.get(k, 0) * kw / 1e9, w.get(k, 0) * kw / a[1]))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2], sys.argv[3])
