#!/usr/bin/env python3
"""The measured tables, generated from a committed bench record so that the prose cannot drift from it:
    python tools/design_table.py profiles/r06_bench.json            # print them
    python tools/design_table.py profiles/r06_bench.json --write    # profiles/MEASURED.md: the per-kernel table and the SQ-counter table (between
                                                                    # their markers); DESIGN.md: the ten-row summary of section 0
"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- bench-table:begin (tools/design_table.py) -->", "<!-- bench-table:end -->"


def f(x, d=3):
    return ("%%.%df" % d) % x


def e(x):
    m = "%.2e" % x
    mant, ex = m.split("e")
    return "%se%d" % (mant, int(ex))


def km(leg):
    """the leg's launch duration: the trimmed mean of round 5's lines, the mean of earlier ones"""
    return leg.get("kernel_ms", leg.get("kernel_ms_mean"))


def table(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    a, r = d["also"], d["roofline"]
    sb = r["same_box"]
    ts, ft, fe, it, ha, s8 = a["transform_set"], a["fused_from_tiles"], a["front_end_and_sad"], a["intra32"], a["host_api"], a["stream8k"]
    cl, mix = ts["classes"], ts["per_ctu_mixed"]
    rows = []
    add = lambda *c: rows.append("  | " + " | ".join(c) + " |")
    add("kernel", "avg launch", "units/s", "of 8 TB/s", "of this box's stream")
    add("---", "---", "---", "---", "---")
    add("`dct32_lds_kernel<fwd>` 2²⁰ blocks", f(r["kernel_ms_per_launch"]) + " ms", e(d["value"]) + " blocks", "**" + f(r["frac"]) + "**", "**" + f(r["frac_of_same_box_copy"]) + "** copy")
    i = a["dct32_inv"]
    add("`dct32_lds_kernel<inv>`", f(km(i)) + " ms", e(i["value"]), f(i["roofline"]["frac"]), f(i["roofline"]["frac_of_same_box_copy"]) + " copy")
    u = a["dct32_fwd_inv_fused"]
    add("`dct32_fwdinv_kernel` (6144 B/block)", f(km(u)) + " ms", e(u["value"]), f(u["hbm_frac"]), f(u["frac_of_same_box_copy"]) + " copy")
    if "reconstruction_only" in u:
        ro = u["reconstruction_only"]
        add("… reconstruction only (`d_coef = NULL`, 4096 B/block)", f(km(ro)) + " ms", e(ro["value"]), f(ro["hbm_frac"]), f(ro["frac_of_same_box_copy"]) + " copy")
    s = a["satd8x8"]
    add("`satd8x8_dma_kernel` 2²⁴ blocks", f(km(s)) + " ms", e(s["value"]) + " blocks", "**" + f(s["roofline"]["frac"]) + "**", "**" + f(s["roofline"]["frac_of_same_box_read"]) + "** read")
    for fam, tag in (("dct2", "`tr_fwd_small_lds_kernel` DCT-II"), ("dst7", "… DST-VII"), ("dct2_inv", "`tr_inv_small_lds_kernel` DCT-II"), ("dst7_inv", "… DST-VII")):
        c = [cl["%s_%dx%d" % (fam, n, n)] for n in (4, 8, 16)]
        add(tag + " 4×4 / 8×8 / 16×16", " / ".join(f(km(x)) for x in c) + " ms", " / ".join(e(x["value"]) for x in c), " / ".join(f(x["hbm_frac"], 2) for x in c),
            " / ".join(f(x["frac_of_same_box_copy"], 2) for x in c) + " copy")
    o, oi, sv = mix["per_ctu_one_launch"], mix["per_ctu_one_launch_inverse"], mix.get("seven_calls_over_offset_tables")
    add("`tr_tiles_kernel` (configs[3], one launch) fwd / inv", f(km(o)) + " / " + f(km(oi)) + " ms", e(o["value"]) + " / " + e(oi["value"]) + " CTUs",
        f(o["hbm_frac"]) + " / " + f(oi["hbm_frac"]), f(o["frac_of_same_box_copy"]) + " / " + f(oi["frac_of_same_box_copy"]) + " copy")
    if sv:
        add("seven calls over offset tables (comparison only)", f(km(sv)) + " ms", e(sv["value"]) + " CTUs", f(sv["hbm_frac"]), f(sv["frac_of_same_box_copy"]))
    dt, st = ft["dct32_from_tiles"], ft["satd8x8_from_tiles"]
    add("`dct32_from_tiles_kernel` / `satd8x8_from_tiles_kernel`", f(km(dt)) + " / " + f(km(st)) + " ms",
        e(dt["value"]) + " / " + e(st["value"]) + " blocks (two-kernel paths: " + e(ft["dct32_residual_then_transform"]["value"]) + " / " + e(ft["satd8x8_residual_then_cost"]["value"]) + ")",
        f(dt["hbm_frac"]) + " / " + f(st["hbm_frac"]), f(dt["frac_of_same_box_copy"]) + " copy / " + f(st["frac_of_same_box_read"]) + " read")
    if "chroma_dct32_from_tiles" in ft:                                  # round 6: the chroma half of the tile stage
        cd, cs, c32, c8 = ft["chroma_dct32_from_tiles"], ft["chroma_satd8x8_from_tiles"], ft["residual_chroma_32"], ft["residual_chroma_8"]
        add("`dct32_chroma_from_tiles_kernel` / `satd8x8_chroma_from_tiles_kernel`", f(km(cd)) + " / " + f(km(cs)) + " ms", e(cd["value"]) + " / " + e(cs["value"]) + " blocks",
            f(cd["hbm_frac"]) + " / " + f(cs["hbm_frac"]), f(cd["frac_of_same_box_copy"]) + " copy / " + f(cs["frac_of_same_box_read"]) + " read")
        if "ctu_dct32_from_tiles" in ft:
            cu = ft["ctu_dct32_from_tiles"]
            add("`dct32_ctu_from_tiles_kernel` (a 4:2:0 CTU's six blocks, CTU order, one launch)", f(km(cu)) + " ms", e(cu["value"]) + " blocks", f(cu["hbm_frac"]), f(cu["frac_of_same_box_copy"]) + " copy")
        add("`residual_chroma_kernel` 32×32 / 8×8 order", f(km(c32)) + " / " + f(km(c8)) + " ms", e(c32["value"]) + " / " + e(c8["value"]) + " blocks",
            f(c32["hbm_frac"]) + " / " + f(c8["hbm_frac"]), f(c32["frac_of_same_box_copy"]) + " / " + f(c8["frac_of_same_box_copy"]) + " copy")
    cv = [fe[k] for k in ("conv_input_fmt", "conv_output_420", "residual_luma_32")]
    add("`tile_convert_kernel` in / out, `residual_luma_kernel`", " / ".join(f(km(x)) for x in cv) + " ms", " / ".join(f(x["GBps"] / 1e3, 2) for x in cv) + " TB/s",
        " / ".join(f(x["hbm_frac"], 2) for x in cv), " / ".join(f(x["frac_of_same_box_copy"], 2) for x in cv) + " copy")
    sd = [fe[k] for k in ("sad_8x8", "sad_16x16", "sad_64x64")]
    add("`sad_kernel` 8×8 / 16×16 / 64×64", " / ".join(f(km(x)) for x in sd) + " ms", " / ".join(f(x["GBps"] / 1e3, 2) for x in sd) + " TB/s",
        " / ".join(f(x["hbm_frac"], 2) for x in sd), " / ".join(f(x["frac_of_same_box_read"], 2) for x in sd) + " read")
    m, m2 = a["satd8x8_me_search"], a["sad8x8_me_search"]
    add("`satd_search_kernel` one 4K frame, ±64", f(km(m)) + " ms", e(m["value"]) + " SATD",
        f(m["frac_of_v_sad_u16_floor"]) + " of the `v_sad_u16` floor at 2.4 GHz, **" + f(m["frac_of_v_sad_u16_floor_at_sclk"]) + "** at the %d MHz sysfs showed during the leg" % m["sclk_mhz"], "–")
    add("`sad_search_kernel`", f(km(m2)) + " ms", e(m2["value"]) + " SAD",
        f(m2["frac_of_v_sad_u8_floor"]) + " of the `v_sad_u8` floor at 2.4 GHz, **" + f(m2["frac_of_v_sad_u8_floor_at_sclk"]) + "** at %d MHz" % m2["sclk_mhz"], "–")
    p, dc = it["predict"], it["decide_35_modes"]
    add("`intra32_predict_kernel` 2.1e6 predictions", f(km(p)) + " ms", e(p["value"]) + " predictions", f(p["written_hbm_frac"]) + " written", f(p["frac_of_same_box_write"]) + " write — NOT write-bound (§11)")
    add("`intra32_costs_kernel` × 35 modes", f(km(dc)) + " ms", e(dc["value"]) + " blocks = " + e(dc["satd8x8_per_s"]) + " SATD", "VALU-bound", "–")
    if "predict_residual_dct32" in it:
        pr = it["predict_residual_dct32"]
        add("`intra32_residual_dct32_kernel` (predict → residual → DCT32, 3072 B/block)", f(km(pr)) + " ms", e(pr["value"]) + " blocks", f(pr["hbm_frac"]), f(pr["frac_of_same_box_copy"]) + " copy — bound by its 1 : 2 read : write mix (§11)")
    frame_bytes = 2 * (32400 * 2048) + 518400 * 128 + 518400 * 4
    add("node layer, one rank, 7680×4320 stream", "%.1f µs per frame (kernel %.1f)" % (s8["ms_per_frame"] * 1e3, s8["kernel_us"]),
        e(s8["frames_per_s"]) + " frames = " + e(s8["dct32_blocks_per_s"]) + " DCT32 + " + e(s8["satd8x8_blocks_per_s"]) + " SATD blocks",
        f(frame_bytes * s8["frames_per_s"] / 8e12, 2), f(frame_bytes * s8["frames_per_s"] / (sb["copy_TBps"] * 1e12), 2) + " copy")
    lk = ha["link_GBps"]
    add("host-pointer `xDct32FwdBatch`, 2¹⁷ blocks, pageable / pinned", f(ha["pageable"]["ms"], 2) + " / " + f(ha["pinned"]["ms"], 2) + " ms", e(ha["pageable"]["blocks_per_s"]) + " / " + e(ha["pinned"]["blocks_per_s"]) + " blocks",
        "PCIe: %.1f / %.1f GB/s each way" % (ha["pageable"]["GBps_each_way"], ha["pinned"]["GBps_each_way"]),
        "%.2f / %.2f of the link with both directions running (%.1f GB/s each way; %.1f / %.1f alone)" % (ha["pageable"]["frac_of_link_both_directions"], ha["pinned"]["frac_of_link_both_directions"],
                                                                                                  lk["each_way_both_directions_at_once"], lk["h2d_alone"], lk["d2h_alone"]))
    if "autotune" in a:                                                  # round 6: default shape / the shape this box's first large call kept
        at = a["autotune"]
        ks = [k for k in ("dct32_fwd_inv_fused", "dct32_reconstruction_only", "satd8x8", "sad_8x8") if k in at]
        add("`\"autotune\"` option: " + " / ".join(k.replace("dct32_", "") for k in ks), "candidate kept: " + " / ".join(str(at[k]["choice"]) for k in ks), "–",
            "default " + " / ".join(f(at[k]["default_hbm_frac"]) for k in ks), "autotuned " + " / ".join(f(at[k]["autotuned_hbm_frac"]) for k in ks) + " (of 8 TB/s)")
    head = ("  This box's streams (`roofline.same_box`): copy %.2f, read %.2f, read-no-store %.2f, write %.2f TB/s.  CPU baseline in the same line: %s blocks/s on %d threads (`%s`).\n"
            % (sb["copy_TBps"], sb["read_TBps"], sb["read_no_store_TBps"], sb["write_TBps"], e(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))
    return head + "\n" + "\n".join(rows)


SUM_BEGIN, SUM_END = "<!-- summary:begin (tools/design_table.py) -->", "<!-- summary:end -->"


def summary(path):
    """ten rows of numbers, every line under 120 columns"""
    d = json.loads(open(path).read().strip().splitlines()[-1])
    a, r = d["also"], d["roofline"]
    ft, it, s8, mix = a["fused_from_tiles"], a["intra32"], a["stream8k"], a["transform_set"]["per_ctu_mixed"]
    u, sat, m = a["dct32_fwd_inv_fused"], a["satd8x8"], a["satd8x8_me_search"]
    rows = ["| leg (BASELINE config) | rate | of 8 TB/s | of this box's stream |", "|---|---|---|---|"]
    add = lambda *c: rows.append("| " + " | ".join(c) + " |")
    add("forward DCT32, 2^20 blocks [1] (headline)", e(d["value"]) + " blocks/s", "**" + f(r["frac"]) + "**", f(r["frac_of_same_box_copy"]) + " copy")
    add("inverse DCT32 [1]", e(a["dct32_inv"]["value"]), f(a["dct32_inv"]["roofline"]["frac"]), f(a["dct32_inv"]["roofline"]["frac_of_same_box_copy"]) + " copy")
    add("fused forward + inverse [1], 6144 B/block", e(u["value"]), f(u["hbm_frac"]), f(u["frac_of_same_box_copy"]) + " copy")
    add("8x8 SATD batch, 2^24 blocks (secondary)", e(sat["value"]) + " blocks/s", "**" + f(sat["roofline"]["frac"]) + "**", f(sat["roofline"]["frac_of_same_box_read"]) + " read")
    add("full-search SATD ME, 4K frame, +-64 [2]", "%.2f ms, %s SATD/s" % (km(m), e(m["value"])), f(m["frac_of_v_sad_u16_floor"]) + " of the VALU floor", "-")
    o = mix["per_ctu_one_launch"]
    add("mixed transform set, CTU order, one launch [3]", e(o["value"]) + " CTUs/s", f(o["hbm_frac"]), f(o["frac_of_same_box_copy"]) + " copy")
    add("7680x4320 stream, node layer, ONE rank [4]", e(s8["frames_per_s"]) + " frames/s", "-", "-")
    dt, st = ft["dct32_from_tiles"], ft["satd8x8_from_tiles"]
    add("tiles -> DCT32 / SATD, luma (f2)", e(dt["value"]) + " / " + e(st["value"]), f(dt["hbm_frac"]) + " / " + f(st["hbm_frac"]),
        f(dt["frac_of_same_box_copy"]) + " copy / " + f(st["frac_of_same_box_read"]) + " read")
    if "chroma_dct32_from_tiles" in ft:
        cd, cs = ft["chroma_dct32_from_tiles"], ft["chroma_satd8x8_from_tiles"]
        add("tiles -> DCT32 / SATD, chroma (f2)", e(cd["value"]) + " / " + e(cs["value"]), f(cd["hbm_frac"]) + " / " + f(cs["hbm_frac"]),
            f(cd["frac_of_same_box_copy"]) + " copy / " + f(cs["frac_of_same_box_read"]) + " read")
    pr = it["predict_residual_dct32"]
    add("intra predict -> residual -> DCT32, one kernel (f4)", e(pr["value"]), f(pr["hbm_frac"]), f(pr["frac_of_same_box_copy"]) + " copy")
    c = d["cpu_baseline"]
    add("reference C (`src_tb/dct32.c`) on %d host threads" % c["cores"], e(c["value"]) + " blocks/s", "-", "bit-exact with the GPU batch")
    out = "\n".join(rows)
    assert max(len(l) for l in rows) <= 120, max(len(l) for l in rows)
    return out


SQ_BEGIN, SQ_END = "<!-- sq-table:begin (tools/design_table.py) -->", "<!-- sq-table:end -->"
SQ_ROWS = [  # (label, kernels of the derived section of <round>_pmc_sq_counters.csv)
    ("`dct32_lds_kernel` fwd / inv", ["dct32_lds_kernel<false>", "dct32_lds_kernel<true>"]),
    ("`dct32_fwdinv_kernel` with / without the coefficient output", ["dct32_fwdinv_kernel<2, true>", "dct32_fwdinv_kernel<2, false>"]),
    ("`satd8x8_dma_kernel` (round 3's staged kernel: 17.7 / 57 / 14 / 66)", ["satd8x8_dma_kernel"]),
    ("`dct32_from_tiles_kernel` / `satd8x8_from_tiles_dma_kernel`", ["dct32_from_tiles_kernel", "satd8x8_from_tiles_dma_kernel"]),
    ("`dct32_chroma_from_tiles_kernel` / `satd8x8_chroma_from_tiles_kernel` / `residual_chroma_kernel` 32 / 8", ["dct32_chroma_from_tiles_kernel", "satd8x8_chroma_from_tiles_kernel", "residual_chroma_kernel<5>", "residual_chroma_kernel<3>"]),
    ("`tr_fwd_small_lds_kernel` 4×4 / 8×8 / 16×16", ["tr_fwd_small_lds_kernel<2>", "tr_fwd_small_lds_kernel<3>", "tr_fwd_small_lds_kernel<4>"]),
    ("`tr_inv_small_lds_kernel` 4×4 / 8×8 / 16×16", ["tr_inv_small_lds_kernel<2>", "tr_inv_small_lds_kernel<3>", "tr_inv_small_lds_kernel<4>"]),
    ("`tr_tiles_kernel` forward / inverse", ["tr_tiles_kernel<false>", "tr_tiles_kernel<true>"]),
    ("`satd_search_kernel` / `sad_search_kernel`", ["satd_search_kernel<8, false, 512, 4>", "sad_search_kernel<2, false>"]),
    ("`intra32_predict_kernel` / `intra32_costs_kernel` / `intra32_residual_dct32_kernel`", ["intra32_predict_kernel", "intra32_costs_kernel", "intra32_residual_dct32_kernel<4>"]),
    ("`sad_kernel` 8×8 / 16×16 / 64×64", ["sad_kernel<2, 4>", "sad_kernel<4, 4>", "sad_kernel<8, 4>"]),
    ("`mem_ceiling_kernel` copy / read / write (the streams)", ["mem_ceiling_kernel<0, 2>", "mem_ceiling_kernel<1, 4>", "mem_ceiling_kernel<2, 2>"]),
]


def sq_table(path):
    """The derived section of profiles/<round>_pmc_sq_counters.csv (tools/collect_profiles.py) as DESIGN.md's SQ table."""
    lines = open(path).read().splitlines()
    at = next(i for i, l in enumerate(lines) if l.startswith("kernel,mfma_busy_frac"))
    rows = {}
    for l in lines[at + 1:]:
        r = l.rsplit(",", 5)                                          # kernel names contain commas
        if len(r) == 6:
            rows[r[0]] = r

    def col(ks, i):
        vals = [float(rows[k][i]) for k in ks]
        return "–" if all(v == 0 for v in vals) else " / ".join("–" if v == 0 else "%.0f" % (100 * v) if i != 1 else "%.1f" % (100 * v) for v in vals) + " %"
    out = ["  | kernel | MFMA busy | VALU busy | LDS busy | wave time waiting |", "  | --- | --- | --- | --- | --- |"]
    for label, ks in SQ_ROWS:
        if "chroma" in label and not all(k in rows for k in ks):       # rounds before 6 have no chroma kernels
            continue
        if not all(k in rows for k in ks):
            raise SystemExit("kernel missing from %s: %s" % (path, [k for k in ks if k not in rows]))
        out.append("  | %s | %s | %s | %s | %s |" % (label, col(ks, 1), col(ks, 2), col(ks, 3), col(ks, 5)))
    return "\n".join(out)


def replace_region(s, begin, end, body, src):
    b, en = s.index(begin), s.index(end)
    return s[:b] + begin + "\n" + "  (generated from `%s`)\n\n" % src + body + "\n\n  " + s[en:]


def main():
    path = sys.argv[1]
    t = table(path)
    sq_path = os.path.join(os.path.dirname(os.path.abspath(path)), os.path.basename(path).split("_")[0] + "_pmc_sq_counters.csv")
    sq = sq_table(sq_path) if os.path.exists(sq_path) else None
    if "--write" not in sys.argv:
        print(t)
        if sq:
            print()
            print(sq)
        return
    p = os.path.join(ROOT, "profiles", "MEASURED.md")
    s = open(p).read()
    s = replace_region(s, BEGIN, END, t, os.path.relpath(os.path.abspath(path), ROOT))
    if sq and SQ_BEGIN in s:
        s = replace_region(s, SQ_BEGIN, SQ_END, sq, os.path.relpath(sq_path, ROOT))
    open(p, "w").write(s)
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    b, en = s.index(SUM_BEGIN), s.index(SUM_END)
    s = s[:b] + SUM_BEGIN + "\n(generated from `%s`)\n\n" % os.path.relpath(os.path.abspath(path), ROOT) + summary(path) + "\n\n" + s[en:]
    open(p, "w").write(s)
    print("profiles/MEASURED.md and DESIGN.md section 0 regenerated from", path)


if __name__ == "__main__":
    main()
