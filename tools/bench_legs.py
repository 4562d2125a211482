"""bench.py's legs other than the headline: each function measures one family the way Bench.timed_leg does and returns its sub-dict of
the FULL record (bench_full.json, under "also"); the compact line keeps one scalar per leg.  `b` is bench.Bench."""
import ctypes
import os
import statistics
import sys
import threading
import time

import numpy as np

from bench_cpu import cpu_baseline_satd

HBM_PEAK_BYTES_PER_S = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
XGMI_LINK_BYTES_PER_S = 153e9          # per direction, per peer link
DCT_BYTES_PER_BLOCK = 4096             # 2048 read + 2048 written   
SATD_BYTES_PER_BLOCK = 132             # 128 read + 4 written
DCT_SEED, SATD_SEED = 0x266, 0x267
P = ctypes.c_void_p


class SclkSampler:
    """median shader clock (MHz) of the device while a leg runs, read from its hwmon freq1_input every 10 ms by a thread;
    None where sysfs does not show it.  The VALU floors of the motion searches scale with it."""

    def __init__(self, sysfs_dir):
        import glob
        files = glob.glob(sysfs_dir + "/hwmon/hwmon*/freq1_input") if sysfs_dir else []
        self.path = files[0] if files else None
        self.samples = []

    def __enter__(self):
        self.stop = False
        if self.path:
            def run():
                while not self.stop:
                    try:
                        self.samples.append(int(open(self.path).read()) / 1e6)
                    except (OSError, ValueError):
                        pass
                    time.sleep(0.01)
            self.thread = threading.Thread(target=run, daemon=True)
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        if self.path:
            self.thread.join()

    def mhz(self):
        busy = [v for v in self.samples if v > 600]                      # idle samples between launches are not the kernel's clock
        return statistics.median(busy) if busy else None


def pcie_link_facts(sysfs_dir):
    """negotiated generation / width of the GPU's PCIe link from sysfs"""
    if not sysfs_dir:
        return None
    try:
        return {"speed": open(sysfs_dir + "/current_link_speed").read().strip(), "width": open(sysfs_dir + "/current_link_width").read().strip(),
                "device": os.path.realpath(sysfs_dir).split("/")[-1]}
    except OSError:
        return None


def smooth_frame_pair(w, h, pad, seed, mv=(5, -3)):
    """(cur [h, w], padded reference [h + 2 pad, w + 2 pad]) uint8: low-passed noise, the reference displaced by `mv` (planted motion)"""
    rs = np.random.RandomState(seed)
    big = rs.randint(0, 256, (h + 2 * pad + 16, w + 2 * pad + 16)).astype(np.float32)
    c = np.cumsum(np.pad(big, ((3, 2), (3, 2)), mode="edge"), axis=0)
    c = c[5:] - c[:-5]
    c = np.cumsum(c, axis=1)
    sm = (c[:, 5:] - c[:, :-5]) / 25.0                                   # 5x5 box low-pass so that motion is findable
    sm = np.clip((sm - 128.0) * 3.0 + 128.0, 0, 255).astype(np.uint8)
    cur = np.ascontiguousarray(sm[pad + 8:pad + 8 + h, pad + 8:pad + 8 + w])
    refp = np.ascontiguousarray(sm[8 - mv[1]:8 - mv[1] + h + 2 * pad, 8 - mv[0]:8 - mv[0] + w + 2 * pad])
    return cur, refp


def leg_dct32_inverse_and_fused(b, x, z, pmc, pmc_src):
    """inverse DCT32 of the headline's coefficients, and coefficients + reconstruction from one pass (6144 B per block)"""
    n, codec, out = b.n_dct, b.codec, {}
    r = b.dev(n * 2048)
    leg = b.timed_leg(lambda: codec.dct32_inv_dev(z.ptr, r.ptr, n, b.stream))
    xs, rs = x.download(np.int16, 4096 * 1024).astype(np.int32), r.download(np.int16, 4096 * 1024).astype(np.int32)
    out["dct32_inv"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg),
                            roofline=b.roofline(leg, DCT_BYTES_PER_BLOCK, n, pmc.get("dct32_inv_bytes_per_launch"), pmc_src),
                            parity="unpinned (no inverse in the reference); bit-exact vs this repo's oracle",
                            roundtrip_max_abs_err=int(np.abs(rs - xs).max()))
    z2 = b.dev(n * 2048)
    leg = b.timed_leg(lambda: codec.dct32_fwd_inv_dev(x.ptr, z2.ptr, r.ptr, n, b.stream))
    out["dct32_fwd_inv_fused"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg), **b.hbm(leg, 6144.0 * n),
                                      same_bytes_as_two_kernels=b.same_on_device(z2, z, n * 2048),
                                      note="2 KiB in, 2 + 2 KiB out per block; inverse fed from the forward's registers, inputs by LDS-DMA (DESIGN.md 3.7)")
    leg = b.timed_leg(lambda: codec.dct32_fwd_inv_dev(x.ptr, 0, r.ptr, n, b.stream), steps=max(4, b.K // 4), warmup=3)
    out["dct32_fwd_inv_fused"]["reconstruction_only"] = dict(value=b.rate(leg, n), unit="blocks/s", **b.brief(leg), **b.hbm(leg, 4096.0 * n),
                                                             note="d_coef = NULL: 2 KiB in, 2 KiB out, the natural-orientation pass 2 is skipped")
    return out


def leg_autotuned(b, x, z):
    """The opt-in "autotune" option (include/x266hip.h): the families whose fastest launch shape differs from box to box, default shape and the shape
    this box's first large call kept, alternating on the same buffers.  The headline and every other leg of the line run the DEFAULTS."""
    codec, n = b.codec, b.n_dct
    r, z2 = b.dev(n * 2048), b.dev(n * 2048)
    ns = b.n_satd
    d, s = b.dev(ns * 128), b.dev(ns * 4)
    codec.fill_residual_dev(d.ptr, ns * 64, SATD_SEED, 0, b.stream)
    nsad = ns * 128 // 2 // 64
    legs = (("dct32_fwd_inv_fused", 6144.0 * n, lambda: codec.dct32_fwd_inv_dev(x.ptr, z2.ptr, r.ptr, n, b.stream)),
            ("dct32_reconstruction_only", 4096.0 * n, lambda: codec.dct32_fwd_inv_dev(x.ptr, 0, r.ptr, n, b.stream)),
            ("satd8x8", float(SATD_BYTES_PER_BLOCK) * ns, lambda: codec.satd8x8_dev(d.ptr, s.ptr, ns, b.stream)),
            ("sad_8x8", 132.0 * nsad, lambda: codec.sad_dev(8, d.ptr, d.ptr + ns * 64, s.ptr, nsad, b.stream)))
    out, short = {}, max(4, b.K // 4)
    for name, nbytes, fn in legs:
        fr = {}
        for mode in (0, 1, 0, 1):                                        # default, tuned, default, tuned: the better of two per mode
            codec.set_option("autotune", mode)
            leg = b.timed_leg(fn, steps=short, warmup=3)
            f = nbytes / (leg["kernel_ms"] * 1e-3) / HBM_PEAK_BYTES_PER_S
            fr[mode] = max(fr.get(mode, 0.0), f)
        out[name] = {"default_hbm_frac": fr[0], "autotuned_hbm_frac": fr[1]}
    codec.set_option("autotune", 0)
    rep = codec.autotune_report()
    names = {"dct32_fwd_inv_fused": "dct32_fwd_inv", "dct32_reconstruction_only": "dct32_recon_only", "satd8x8": "satd8x8", "sad_8x8": "sad8"}
    for name, fam in names.items():
        out[name].update(choice=rep.get(fam, {}).get("choice"), candidate_ms=rep.get(fam, {}).get("ms"))
    out["same_bytes_default_and_tuned"] = b.same_on_device(z2, z, n * 2048)
    out["note"] = "candidate 0 = the default shape; candidates in x266hip_abi.hip (kFwdInvCands, kReconCands, kSatdCands, kSadCands)"
    return out


def leg_satd(b, pmc, pmc_src):
    """the 8x8 SATD residual batch (the secondary metric), with the port timed on the host cores at N = 1"""
    n, codec = b.n_satd, b.codec
    d, s = b.dev(n * 128), b.dev(n * 4)
    codec.fill_residual_dev(d.ptr, n * 64, SATD_SEED, b.rank * n * 64, b.stream)
    leg = b.timed_leg(lambda: codec.satd8x8_dev(d.ptr, s.ptr, n, b.stream))
    out = dict(value=b.rate(leg, n), unit="blocks/s", blocks_per_gpu=n, **b.brief(leg),
               roofline=b.roofline(leg, SATD_BYTES_PER_BLOCK, n, pmc.get("satd8x8_bytes_per_launch"), pmc_src, box_kind="read"))
    if b.rank == 0 and b.world == 1 and not b.args.no_cpu_baseline:
        ns = min(n, 1 << 23)
        out["cpu_baseline"] = cpu_baseline_satd(d.download(np.int16, ns * 64).reshape(ns, 64), s.download(np.uint32, ns))
    return out


def leg_motion_search(b, keep):
    """BASELINE configs[2]: full-search motion estimation of one 3840x2160 luma frame, window +-64 -- SATD, then the SAD metric"""
    codec = b.codec
    w, h, rng = 3840, 2160, 64
    cur_h, refp_h = smooth_frame_pair(w, h, rng, 0x266 + b.rank)           # planted motion (5, -3)
    cur, refp = b.dev_from(cur_h), b.dev_from(refp_h)
    rstride = refp_h.shape[1]
    nb = (w // 8) * (h // 8)
    best = b.dev(nb * 8)
    origin = refp.ptr + rng * rstride + rng
    keep.update(cur=cur, refp=refp, best=best, origin=origin, rstride=rstride)
    ncand = nb * (2 * rng + 1) ** 2
    me_steps = max(4, b.K // 4)
    out = {}
    # VALU floors: 32 x v_sad_u16 (SATD) / 16 x v_sad_u8 (SAD) per 64 candidates, 4 cycles per wave64 instruction (tools/probes/alubench), at the
    # 2.4 GHz the part is specified for and -- where sysfs shows it -- at the shader clock this box sustained during the leg
    for key, fn, per64, unit, floor_name, steps in (("satd8x8_me_search", codec.satd_search_dev, 32, "SATD/s", "v_sad_u16", max(me_steps, 40)),
                                                    ("sad8x8_me_search", codec.sad_search_dev, 16, "SAD/s", "v_sad_u8", max(me_steps, 60))):
        with SclkSampler(b.sysfs) as clk:
            leg = b.timed_leg(lambda f=fn: f(cur.ptr, w, origin, rstride, w, h, rng, best.ptr, 0, b.stream), steps=steps, warmup=2)
        mv = best.download(np.int16, nb * 4).reshape(nb, 4)[:, :2]
        cycles = ncand / 64 * per64 * 4 / (4 * b.info["cu_count"])
        sclk = clk.mhz()
        out[key] = dict(value=b.rate(leg, ncand), unit=unit, ms_per_frame=leg["ms_per_step"], **b.brief(leg), sclk_mhz=sclk,
                        planted_mv_found_fraction=float(((mv[:, 0] == 5) & (mv[:, 1] == -3)).mean()))
        out[key]["frac_of_%s_floor" % floor_name] = cycles / 2.4e9 / (leg["kernel_ms"] * 1e-3)
        out[key]["frac_of_%s_floor_at_sclk" % floor_name] = (cycles / (sclk * 1e6) / (leg["kernel_ms"] * 1e-3)) if sclk else None
    out["satd8x8_me_search"].update(
        frame="%dx%d luma, 8x8 blocks, window +-%d (%d candidates per block)" % (w, h, rng, (2 * rng + 1) ** 2),
        bound="VALU issue (v_sad_u16), not HBM: ~18 MB of compulsory traffic per frame",
        parity="per-candidate cost pinned by satd8x8 (src_tb/satd.c); harness (order, tie-break, padding) unpinned")
    out["sad8x8_me_search"]["parity"] = "metric = sad() of riscv/programs/benchmarks/sad/sad.c at n = 8; harness unpinned, as for the SATD search"
    return out


def leg_transform_set(b, x):
    """BASELINE configs[3]: the mixed transform set (DCT-II 4..32 + closed-form DST-VII 4/8/16), 2 GiB of residual per class,
    and the CTU-ordered mixed buffer in one launch"""
    codec, n_dct = b.codec, b.n_dct
    zt = b.dev(n_dct * 2048)                                            # own output buffer: z still holds the headline leg's result
    short = max(4, b.K // 4)
    ts = {}
    for ttype, tname, inverse in ((0, "dct2", False), (1, "dst7", False), (0, "dct2_inv", True), (1, "dst7_inv", True)):
        for n in (4, 8, 16):
            nblk = (n_dct * 1024) // (n * n)
            call = codec.transform_inv_dev if inverse else codec.transform_fwd_dev
            leg = b.timed_leg(lambda c=call, tt=ttype, nn=n, cnt=nblk: c(tt, nn, x.ptr, zt.ptr, cnt, 0, b.stream), steps=short, warmup=3)
            ts["%s_%dx%d" % (tname, n, n)] = dict(value=b.rate(leg, nblk), unit="blocks/s", **b.hbm(leg, 4.0 * n * n * nblk), **b.brief(leg))
    # per-CTU mixed batch: every 64x64 CTU's 32x32 quadrants cycle through the seven (type, size) classes; every quadrant is a tile
    # with its own class byte, the whole buffer is ONE launch (xTransformTilesDev)
    n_ctu = (n_dct * 1024) // 4096
    q = np.arange(n_ctu * 4, dtype=np.int64)
    tile_cls = b.dev_from(np.array([3, 2, 6, 1, 5, 0, 4], np.uint8)[(q + q // 4) % 7])   # kinds -> type*4 + log2N-2
    per_ctu = {"layout": "64x64 CTUs whose 32x32 quadrants cycle through the seven classes (DCT-II 32/16/8/4, DST-VII 16/8/4), "
                         "TUs of a quadrant contiguous", "ctus": n_ctu}
    for inv_flag, name in ((0, "per_ctu_one_launch"), (1, "per_ctu_one_launch_inverse")):
        leg = b.timed_leg(lambda f=inv_flag: codec.transform_tiles_dev(f, x.ptr, zt.ptr, n_ctu * 4, 0, tile_cls.ptr, b.stream), steps=short, warmup=3)
        per_ctu[name] = dict(value=b.rate(leg, n_ctu), unit="CTUs/s", **b.hbm(leg, 4.0 * n_ctu * 4096), **b.brief(leg))
    return {"classes": ts, "per_ctu_mixed": per_ctu, "parity": "unpinned upstream except DCT-II 32; bit-exact vs this repo's oracle",
            "note": "4*N*N algorithmic bytes per block; hbm_frac from the trimmed mean of the timed launches' HIP-event durations; the seven-calls-over-offset-"
                    "tables form of the mixed buffer (rounds 1-4: 0.59 of 8 TB/s, superseded by the one-launch call) is still tested, no longer benched"}


def leg_fused_from_tiles(b):
    """Tiled cur / pred frames -> coefficients / costs, residual never in HBM, next to the two-kernel paths.  32768^2 luma: exactly 2^20
    DCT32 blocks and 2^24 SATD blocks, i.e. the two-kernel legs launch the headline kernels at the headline sizes."""
    codec = b.codec
    fw, fh = 32768, 32768
    ntile = (fw // 16) * (fh // 16)
    tcur, tpred = b.dev_random_bytes(ntile * 512, 0x266), b.dev_random_bytes(ntile * 512, 0x268)
    fcoef, fcost, fres = b.dev(fw * fh * 2), b.dev(fw * fh // 64 * 4), b.dev(fw * fh * 2)
    short = max(4, b.K // 4)
    fused = {}
    legs = (("dct32_from_tiles", fw * fh // 1024, 4096, lambda: codec.dct32_fwd_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcoef.ptr, b.stream)),
            ("dct32_residual_then_transform", fw * fh // 1024, None,
             lambda: (codec.residual_luma_dev(tcur.ptr, tpred.ptr, fw, fh, 32, fres.ptr, b.stream), codec.dct32_fwd_dev(fres.ptr, fcoef.ptr, fw * fh // 1024, b.stream))),
            ("satd8x8_from_tiles", fw * fh // 64, 132, lambda: codec.satd8x8_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcost.ptr, b.stream)),
            ("satd8x8_residual_then_cost", fw * fh // 64, None,
             lambda: (codec.residual_luma_dev(tcur.ptr, tpred.ptr, fw, fh, 8, fres.ptr, b.stream), codec.satd8x8_dev(fres.ptr, fcost.ptr, fw * fh // 64, b.stream))))
    # the chroma half (m_C of the same tiles, src/x266.cpp:60): per 64x64 CTU one 32x32 U and one 32x32 V block -> 2^19 DCT32 blocks; per tile
    # one 8x8 U and V block -> 2^23 SATD blocks.  Planar U / V output streams.  The read side touches ONE 128-byte line of every 512-byte tile.
    nc32, nc8 = fw * fh // 4096 * 2, ntile * 2
    fctu = b.dev(fw * fh * 3)
    legs += (("chroma_dct32_from_tiles", nc32, 4096,
              lambda: codec.dct32_fwd_chroma_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcoef.ptr, fcoef.ptr + nc32 * 1024, 1, b.stream)),
             ("chroma_satd8x8_from_tiles", nc8, 132,
              lambda: codec.satd8x8_chroma_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fcost.ptr, fcost.ptr + ntile * 4, 1, b.stream)),
             ("ctu_dct32_from_tiles", fw * fh // 4096 * 6, 4096,      # a whole 4:2:0 CTU per 12 KiB of output: Y0 Y1 Y2 Y3 U V, one launch
              lambda: codec.dct32_fwd_ctu_from_tiles_dev(tcur.ptr, tpred.ptr, fw, fh, fctu.ptr, b.stream)),
             ("residual_chroma_32", nc32, 4096,
              lambda: codec.residual_chroma_dev(tcur.ptr, tpred.ptr, fw, fh, 32, fres.ptr, fres.ptr + nc32 * 1024, 1, b.stream)),
             ("residual_chroma_8", nc8, 256,
              lambda: codec.residual_chroma_dev(tcur.ptr, tpred.ptr, fw, fh, 8, fres.ptr, fres.ptr + nc8 * 64, 1, b.stream)))
    for name, units, bytes_per_unit, fn in legs:
        leg = b.timed_leg(fn, steps=short, warmup=3)
        fused[name] = dict(value=b.rate(leg, units), unit="blocks/s", **b.brief(leg))
        if bytes_per_unit:
            fused[name].update(b.hbm(leg, bytes_per_unit * units, "read" if "satd" in name else "copy"))
    fused["note"] = ("%dx%d tiled frame pair (x266.cpp ref_block_t); fused kernels are bit-identical to the two-kernel paths "
                     "listed next to them (tests/test_gpu_tiles.py)" % (fw, fh))
    return fused


def leg_front_end_and_sad(b):
    """frame container conversion, residual formation, SAD -- pure data movement, HBM-bound"""
    codec = b.codec
    fw2, fh2 = 16384, 16384                                              # 256 Mi luma samples
    npx = fw2 * fh2
    ypl = b.dev_random_bytes(npx, 0x77 + b.rank)
    upl, vpl = b.dev_random_bytes(npx // 4, 0x78), b.dev_random_bytes(npx // 4, 0x79)
    t_a, t_b = b.dev_random_bytes(npx * 2, 0x7a), b.dev_random_bytes(npx * 2, 0x7b)   # 512-byte tiles: 2 bytes per luma sample
    res2, sad_o = b.dev(npx * 2), b.dev(npx // 64 * 4)
    short = max(4, b.K // 4)
    front = {}
    for name, nbytes, fn in (
            ("conv_input_fmt", 3.0 * npx, lambda: codec.conv_input_fmt_dev(t_a.ptr, ypl.ptr, upl.ptr, vpl.ptr, fw2, fw2, fh2, b.stream)),
            ("conv_output_420", 3.0 * npx, lambda: codec.conv_output_420_dev(t_a.ptr, ypl.ptr, fw2, upl.ptr, vpl.ptr, fw2 // 2, fw2, fh2, b.stream)),
            ("residual_luma_32", 4.0 * npx, lambda: codec.residual_luma_dev(t_a.ptr, t_b.ptr, fw2, fh2, 32, res2.ptr, b.stream)),
            ("sad_8x8", 2.0 * npx + 4.0 * (npx // 64), lambda: codec.sad_dev(8, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 64, b.stream)),
            ("sad_16x16", 2.0 * npx + 4.0 * (npx // 256), lambda: codec.sad_dev(16, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 256, b.stream)),
            ("sad_64x64", 2.0 * npx + 4.0 * (npx // 4096), lambda: codec.sad_dev(64, ypl.ptr, t_b.ptr, sad_o.ptr, npx // 4096, b.stream))):
        leg = b.timed_leg(fn, steps=short, warmup=3)
        front[name] = dict(GBps=b.world * nbytes * leg["steps"] / leg["wall_s"] / 1e9, **b.hbm(leg, nbytes, "read" if name.startswith("sad") else "copy"),
                           samples_per_s=b.world * npx * leg["steps"] / leg["wall_s"], **b.brief(leg))
    front["note"] = ("%dx%d frame; bytes = planes read + tile bytes written (conv), luma of both tile frames + int16 residual "
                     "(residual), both blocks + 4-byte result (sad)" % (fw2, fh2))
    return front


def leg_intra(b):
    """32x32 intra prediction, mode decision, and prediction -> residual -> DCT32 in one kernel (HEVC 35 modes; parity unpinned upstream)"""
    codec = b.codec
    n_sets = 59918                                                       # x 35 modes = 2 GiB of predictions
    rs = np.random.RandomState(0x32 + b.rank)
    refs_t = b.dev_from(rs.randint(0, 256, (n_sets, 144)).astype(np.uint8))
    modes_t = b.dev_from(np.tile(np.arange(35, dtype=np.uint8), n_sets))
    index_t = b.dev_from(np.repeat(np.arange(n_sets, dtype=np.int32), 35))
    pred_t = b.dev(n_sets * 35 * 1024)
    n_dec = min(1 << 17, n_sets)
    src_t = b.dev_random_bytes(n_dec * 1024, 0x33)
    cost_t, bestm_t = b.dev(n_dec * 35 * 4), b.dev(n_dec)
    short = max(4, b.K // 4)
    intra = {}
    leg = b.timed_leg(lambda: codec.intra32_predict_dev(refs_t.ptr, modes_t.ptr, index_t.ptr, pred_t.ptr, n_sets * 35, b.stream), steps=short, warmup=3)
    written = 1024.0 * n_sets * 35 / (leg["kernel_ms"] * 1e-3)
    intra["predict"] = dict(value=b.rate(leg, n_sets * 35), unit="predictions/s", written_hbm_frac=written / HBM_PEAK_BYTES_PER_S,
                            frac_of_same_box_write=b.of_box(written, "write"), **b.brief(leg))
    leg = b.timed_leg(lambda: codec.intra32_costs_dev(refs_t.ptr, src_t.ptr, cost_t.ptr, bestm_t.ptr, n_dec, b.stream), steps=short, warmup=3)
    intra["decide_35_modes"] = dict(value=b.rate(leg, n_dec), unit="blocks/s", satd8x8_per_s=b.rate(leg, n_dec) * 35 * 16, **b.brief(leg))
    if hasattr(codec, "intra32_residual_dct32_dev"):
        # the encoder loop's form: the chosen mode's prediction never reaches HBM (1 KiB of source + 144 B of references in, 2 KiB of coefficients out)
        n_blk = min(1 << 20, n_sets * 35)
        src_b, coef_b = b.dev_random_bytes(n_blk * 1024, 0x34), b.dev(n_blk * 2048)
        leg = b.timed_leg(lambda: codec.intra32_residual_dct32_dev(refs_t.ptr, modes_t.ptr, index_t.ptr, src_b.ptr, coef_b.ptr, n_blk, b.stream), steps=short, warmup=3)
        intra["predict_residual_dct32"] = dict(value=b.rate(leg, n_blk), unit="blocks/s", **b.hbm(leg, 3072.0 * n_blk), **b.brief(leg),
                                               note="xIntra32ResidualDct32Dev: predict (given mode) -> src - pred -> forward DCT32 in one kernel; 1 KiB in + 2 KiB out per block")
    intra["parity"] = "unpinned upstream (src/mkIntra32-wip.bsv is a sketch without a model); bit-exact vs this repo's oracle"
    return intra


def leg_host_api(b, n):
    """The literal drop-in path: xDct32FwdBatch on n blocks from pageable and from pinned host buffers (what INTEGRATION.md section 2 tells an
    x266.cpp maintainer to call, src/x266.cpp:526-555), next to what the link itself gives (plain copies).  PCIe-inclusive -- never `value`."""
    codec, hip = b.codec, b.hip
    nbytes = n * 2048
    out = {"blocks": n, "MiB_each_way": nbytes >> 20, "pcie_link": pcie_link_facts(b.sysfs)}

    def best_of(fn, reps=4):
        best = 1e9
        for _ in range(reps):
            hip.device_sync()
            t0 = time.perf_counter()
            fn()
            hip.device_sync()
            best = min(best, time.perf_counter() - t0)
        return best
    hp_in, hp_out = codec.host_alloc(nbytes, np.uint8), codec.host_alloc(nbytes, np.uint8)
    d_a, d_b = b.dev(nbytes), b.dev(nbytes)
    streams = [hip.stream_create() for _ in range(4)]
    H2D, D2H = 1, 2                                                      # hipMemcpyHostToDevice / DeviceToHost

    def both(s1, s2):
        hip.memcpy_async(d_a.ptr, hp_in.ctypes.data, nbytes, H2D, s1)
        hip.memcpy_async(hp_out.ctypes.data, d_b.ptr, nbytes, D2H, s2)
    # HIP multiplexes streams onto a few hardware queues; two streams that land on the same one serialise their copies. The link's
    # two-way rate is what the best of a few stream pairs reaches.
    two_way = max(nbytes / best_of(lambda a=a, c=c: both(streams[a], streams[c]), reps=2) / 1e9 for a, c in ((0, 1), (0, 2), (1, 3), (2, 3)))
    out["link_GBps"] = {"h2d_alone": nbytes / best_of(lambda: hip.memcpy_async(d_a.ptr, hp_in.ctypes.data, nbytes, H2D, streams[0])) / 1e9,
                        "d2h_alone": nbytes / best_of(lambda: hip.memcpy_async(hp_out.ctypes.data, d_b.ptr, nbytes, D2H, streams[1])) / 1e9,
                        "each_way_both_directions_at_once": two_way,
                        "how": "one plain %d MiB hipMemcpyAsync from / to pinned memory per direction; two-way: best of four stream pairs" % (nbytes >> 20)}
    for s in streams:
        hip.lib.hipStreamDestroy(s)
    del hp_in, hp_out, d_a, d_b
    x_dev = b.dev(nbytes)
    codec.fill_residual_dev(x_dev.ptr, n * 1024, DCT_SEED, 0, 0)
    xh = x_dev.download(np.int16, n * 1024).reshape(n, 1024)            # pageable, touched
    zh = np.ones_like(xh)

    def call(i, o):
        rc = codec.L.xDct32FwdBatch(codec.ctx, P(i), P(o), n)
        if rc:
            raise RuntimeError("xDct32FwdBatch failed: %d" % rc)
    dt = best_of(lambda: call(xh.ctypes.data, zh.ctypes.data))
    out["pageable"] = {"blocks_per_s": n / dt, "GBps_each_way": nbytes / dt / 1e9, "ms": dt * 1e3}
    xp, zp = codec.host_alloc((n, 1024), np.int16), codec.host_alloc((n, 1024), np.int16)
    xp[:] = xh
    dt = best_of(lambda: call(xp.ctypes.data, zp.ctypes.data))
    out["pinned"] = {"blocks_per_s": n / dt, "GBps_each_way": nbytes / dt / 1e9, "ms": dt * 1e3, "same_result_as_pageable": bool(np.array_equal(zp, zh))}
    for k in ("pinned", "pageable"):
        out[k]["frac_of_link_both_directions"] = out[k]["GBps_each_way"] / two_way
    # one block per call -- what the BDPI shims do (the <= 64 KiB path: the kernel runs on page-locked host memory, one launch + one synchronize)
    x1, z1 = xh[:1].copy(), np.empty((1, 1024), np.int16)
    for _ in range(50):
        codec.L.xDct32FwdBatch(codec.ctx, P(x1.ctypes.data), P(z1.ctypes.data), 1)
    t0 = time.perf_counter()
    for _ in range(1000):
        codec.L.xDct32FwdBatch(codec.ctx, P(x1.ctypes.data), P(z1.ctypes.data), 1)
    out["one_block_call_us"] = (time.perf_counter() - t0) / 1000 * 1e6
    out["one_block_call_same_result"] = bool(np.array_equal(z1[0], zh[0]))
    out["note"] = ("host pointers in and out, best of 4 calls: 16 MiB chunks over three staging slots, uploads + kernels issued by the calling thread, "
                   "downloads by a helper thread (a pageable copy blocks its issuing thread); inputs are NOT resident, so this is never `value`")
    return out


def leg_node_stream8k(b, node, also):
    """BASELINE configs[4]: the 7680x4320 frame stream through the node layer (one RCCL group per step with N > 1; in place with one rank)"""
    codec, rank, world = b.codec, b.rank, b.world
    fw8, fh8 = 7680, 4320
    nd8, ns8 = (fw8 // 32) * (fh8 // 32), (fw8 // 8) * (fh8 // 8)
    IN_RING, OUT_RING = 4, 5                                             # X266_STREAM_IN_RING / X266_STREAM_OUT_RING (include/x266hip.h)
    fin = fout = None
    if rank == 0:
        fin = [(b.dev(nd8 * 2048), b.dev(ns8 * 128)) for _ in range(IN_RING)]
        fout = [(b.dev(nd8 * 2048), b.dev(ns8 * 4)) for _ in range(OUT_RING)]
        for i, (a, d) in enumerate(fin):
            codec.fill_residual_dev(a.ptr, nd8 * 1024, DCT_SEED, i * 100000007, b.stream)
            codec.fill_residual_dev(d.ptr, ns8 * 64, SATD_SEED, i * 100000007, b.stream)
    b.hip.device_sync()
    st8 = node.frame_stream(fw8, fh8)
    # one foreign call per frame: the argument arrays of every (input ring, output ring) pairing are built once
    prep8 = ([st8.prepare([fin[i % IN_RING][0].ptr, fin[i % IN_RING][1].ptr], [fout[i % OUT_RING][0].ptr, fout[i % OUT_RING][1].ptr])
              for i in range(IN_RING * OUT_RING)] if rank == 0 else None)
    raw_next = node.L.xNodeStreamNextSlotStream

    def push8(f):
        if rank == 0:                                                    # resident inputs: "produced" on the frame's own slot stream, so the push needs no producer event
            st8.push_prepared(prep8[f % (IN_RING * OUT_RING)], raw_next(st8.s))
        else:
            st8.push()
    F = b.args.stream8k
    # clocks: ~0.1 s of frames before the timed ones (200 frames are 7 ms); a fixed count, the same on every rank
    for f in range(2500 if world == 1 else 64):
        push8(f)
    st8.flush()
    b.barrier()
    t0 = time.perf_counter()
    for f in range(F):
        push8(f)
    st8.flush()
    b.barrier()
    wall8 = b.max_over_ranks(time.perf_counter() - t0)
    exact8 = kernel_us = None
    if rank == 0:                                                        # last frame against the plain single-device calls
        a, d = fin[(F - 1) % IN_RING]
        c, e = fout[(F - 1) % OUT_RING]
        c1, e1 = b.dev(nd8 * 2048), b.dev(ns8 * 4)
        codec.dct32_fwd_dev(a.ptr, c1.ptr, nd8, b.stream)
        codec.satd8x8_dev(d.ptr, e1.ptr, ns8, b.stream)
        b.hip.device_sync()
        exact8 = b.same_on_device(c, c1, nd8 * 2048) and b.same_on_device(e, e1, ns8 * 4)
    if rank == 0 and world == 1:                                         # what the frame's one launch costs by itself, back to back on one stream
        a, d = fin[0]
        c, e = fout[0]
        for phase in (0, 1):
            for _ in range(200):
                codec.frame_lanes_dev(a.ptr, c.ptr, nd8, d.ptr, e.ptr, ns8, b.stream)
            codec.event_record(b.events[phase], b.stream)
        kernel_us = codec.event_elapsed_ms(b.events[0], b.events[1]) / 200 * 1e3
    link_bytes = (nd8 * 2048 + ns8 * 128) / world                        # one peer's input shard of a frame, over one link
    also["stream8k"] = {
        "frames_per_s": F / wall8, "ms_per_frame": wall8 / F * 1e3, "frames": F,
        "kernel_us": kernel_us, "launches_per_frame_and_rank": 1,
        "kernel_share_of_frame_time": (kernel_us * 1e-6 / (wall8 / F)) if kernel_us else None,
        "dct32_blocks_per_s": nd8 * F / wall8, "satd8x8_blocks_per_s": ns8 * F / wall8,
        "frame": "7680x4320: %d DCT32 blocks (66.4 MB) + %d SATD blocks (66.4 MB)" % (nd8, ns8),
        "path": "C ABI node layer (xNodeStreamPush / Flush): per step one RCCL group carries frame t's shards root -> peers and "
                "frame t-2's coefficients and costs peers -> root on a communication stream while every rank transforms frame t"
                if world > 1 else "C ABI node layer, one rank: the root transforms the frame in place, no transfer (RCCL only in the self-test)",
        "bit_exact_vs_single_device": exact8,
        "link_bound_frames_per_s": (XGMI_LINK_BYTES_PER_S / link_bytes) if world > 1 else None,
        "link_bound": "each peer's input shard crosses ONE xGMI link (~153 GB/s per direction): <= 7.5e7 DCT32 blocks/s per peer (SURVEY.md 8e)"}
    st8.close()


def leg_node_batch_and_search(b, node, also, x, z, me):
    """one resident batch scattered and gathered (SURVEY 8e "end-to-end scatter -> compute -> gather"), and the sharded motion search"""
    from x266_amd.node import OP_DCT32_FWD
    rank, world = b.rank, b.world
    nsg = min(1 << 18, b.n_dct)
    pin, pout = (x.ptr, z.ptr) if rank == 0 else (0, 0)
    b.hip.device_sync()
    node.batch_scatter_gather(OP_DCT32_FWD, pin, pout, nsg, 0)
    b.barrier()
    t0 = time.perf_counter()
    for _ in range(4):
        node.batch_scatter_gather(OP_DCT32_FWD, pin, pout, nsg, 0)
    b.barrier()
    wall_sg = b.max_over_ranks(time.perf_counter() - t0) / 4
    also["dct32_scatter_gather"] = {"value": nsg / wall_sg, "unit": "blocks/s", "blocks": nsg,
                                    "link_bound_blocks_per_s": (world * XGMI_LINK_BYTES_PER_S / 2048.0) if world > 1 else None,
                                    "link_bound": "every peer's shard crosses ONE xGMI link (~153 GB/s per direction, inputs one way, results the other): "
                                                  "<= 153e9 / 2048 = 7.5e7 blocks/s per peer, i.e. world x 7.5e7 with the root computing its own shard in place"
                                                  if world > 1 else None,
                                    "note": "root-resident batch cut into chunks (8 MiB of input per rank), pipelined through the node stream "
                                            "(xNodeBatchScatterGather); at N = 1 no transfer"}
    if not me:
        return
    # sharded motion search: stripes + halo from the root, records back (xNodeSatd8x8Search)
    nstr = max(world, 1)
    cur, origin, best = (me["cur"].ptr, me["origin"], me["best"].ptr) if rank == 0 else (0, 0, 0)
    nb8 = (3840 // 8) * (2160 // 8) * 8

    def search():
        node.satd_search(cur, 3840, origin, me["rstride"], 3840, 2160, 64, nstr, best)
    search()
    ref_best = me["best"].download(np.uint8, nb8) if rank == 0 else None
    b.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        search()
    b.barrier()
    wall_ms = b.max_over_ranks(time.perf_counter() - t0) / 3
    same = None
    if rank == 0:
        b.codec.satd_search_dev(cur, 3840, origin, me["rstride"], 3840, 2160, 64, best, 0, b.stream)
        same = bool(np.array_equal(me["best"].download(np.uint8, nb8), ref_best))
    also["satd8x8_me_search_sharded"] = {"ms_per_frame": wall_ms * 1e3, "stripes": nstr, "identical_to_single_device": same,
                                         "note": "synchronous call incl. scatter of cur stripes + reference halo and gather of (mv, cost)"}


def node_legs(b, also, x, z, me):
    """The node layer of the C ABI: BASELINE configs[4] and the other end-to-end scatter / gather figures -- the only legs that talk RCCL."""
    from x266_amd.node import Node
    if os.environ.get("X266_BENCH_TEST_STALL_RANK") == str(b.rank):         # test hook (never set by the driver): this rank never joins the node layer
        time.sleep(1e6)
    uid = [Node.unique_id() if b.rank == 0 else None]
    if b.dist is not None:
        b.dist.broadcast_object_list(uid, src=0)
    node = Node.for_rank(b.local_rank, b.rank, b.world, uid[0])      # xHipNodeInitRank: one process per GPU, also at N = 1
    node.self_test()                                                    # RCCL ring send/recv + all-reduce, checked
    ver, path = Node.rccl_info()
    infos = ["%s (version %d)" % (path, ver)]
    if b.dist is not None:
        infos = [None] * b.world
        b.dist.all_gather_object(infos, "%s (version %d)" % (path, ver))
    also["rccl_by_rank"] = infos                                        # which library each rank's node layer talks to
    leg_node_stream8k(b, node, also)
    leg_node_batch_and_search(b, node, also, x, z, me)
    node.close()


def run_node_legs_under_watchdog(b, result, also, x, z, me, json_fd, emit):
    """A communication hang must not cost the whole line: past --node-timeout seconds rank 0 prints the JSON with what it has
    (the legs marked as timed out) and every rank leaves."""
    def node_timed_out():
        also["node_layer_error"] = "node-layer legs did not finish within %.0f s (RCCL hang?); line printed without them" % b.args.node_timeout
        if b.rank == 0:
            result.setdefault("cpu_baseline", None)
            result["device"] = b.info["name"].strip()
            emit(result, json_fd)
        os._exit(0)
    watchdog = threading.Timer(b.args.node_timeout, node_timed_out)
    watchdog.daemon = True
    watchdog.start()
    try:
        node_legs(b, also, x, z, me)
    except Exception as e:                                              # e.g. RCCL missing: keep the rest of the line
        also["node_layer_error"] = "%s: %s" % (type(e).__name__, e)
    watchdog.cancel()
