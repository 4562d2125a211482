#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a bench run from gpurun_out/ into profiles/ (tracked).
usage: collect_profiles.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_json> [pytest_log [pmc_sq_dir pmc_lds_dir]]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit("nothing matches " + pattern)
    return hits[-1]                      # the newest: gpurun_out/ keeps earlier rounds' directories of the same name


def short(n):
    n = n.replace(" ", "")
    if "mem_ceiling_kernel<0" in n: return "mem_ceiling_kernel<copy>"            # round 4: the calibration kernel (reads and writes exactly the buffer)
    if "dct32_lds_kernel<true>" in n: return "dct32_kernel<inverse>"              # round 4 names
    if "dct32_lds_kernel<false>" in n: return "dct32_kernel<forward>"
    if "satd8x8_dma_kernel" in n: return "satd8x8_kernel"
    if "dct32_lds_kernel<1" in n or "dct32_kernel<1" in n or "dct32_kernel<true" in n: return "dct32_kernel<inverse>"
    if "dct32_lds_kernel<0" in n or "dct32_kernel<0" in n or "dct32_kernel<false" in n: return "dct32_kernel<forward>"
    if "dct32_kernel<2" in n: return "dct32_kernel<passthrough>"
    if "satd8x8_kernel" in n or "satd8x8_lds_kernel" in n: return "satd8x8_kernel"
    if "fill_residual" in n: return "fill_residual_kernel"
    return None


def kname(n):
    import re
    m = re.search(r"::(\w+(?:<[^>]*>)?)\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[:40]


def sq_summary(tag, sq_dir, lds_dir):
    """SQ / LDS counters -> profiles/<tag>_pmc_sq_counters.csv plus derived per-SIMD utilisations."""
    out = ["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_WAVE_CYCLES "
           "SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE   and a second pass   --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS "
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY   -- python bench.py --no-cpu-baseline --steps 5 --warmup 2",
           "# mean per dispatch, summed over the chip as rocprofv3 reports it", "kernel,counter,dispatches,mean"]
    tab = collections.defaultdict(dict)
    for d in (sq_dir, lds_dir):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
            if "x266" in r["Kernel_Name"]:
                acc[(kname(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            out.append("%s,%s,%d,%.6g" % (k, c, len(v), sum(v) / len(v)))
            tab[k][c] = sum(v) / len(v)
    out.append("# derived: cycles = GRBM_GUI_ACTIVE / 8 XCDs; per-SIMD busy = counter / (1024 SIMDs * cycles); "
               "SQ_ACTIVE_INST_VALU counts quad-cycles; LDS per CU (256)")
    out.append("kernel,mfma_busy_frac,valu_busy_frac,lds_busy_frac,cycles_per_mfma,wave_wait_frac")
    for k, v in sorted(tab.items()):
        if "GRBM_GUI_ACTIVE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        out.append("%s,%.3f,%.3f,%.3f,%.1f,%.2f" % (
            k, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc),
            v.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_INSTS_VALU_MFMA_I8"], 1),
            v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
    open(os.path.join(P, tag + "_pmc_sq_counters.csv"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-12:]))


def main():
    tag, stats_dir, fdir, wdir, bench = sys.argv[1:6]
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_stats.csv")))))
    with open(os.path.join(P, tag + "_bench_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    out = {}
    lines = ["# rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --no-cpu-baseline --steps 5 --warmup 2",
             "# one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass); units: KB per dispatch",
             "kernel,counter,dispatches,mean_KB,min_KB,max_KB"]
    for ctr, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == ctr:
                acc[k].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            lines.append("%s,%s,%d,%.1f,%.1f,%.1f" % (k, ctr, len(v), sum(v) / len(v), min(v), max(v)))
            out[(k, ctr)] = sum(v) / len(v)
    open(os.path.join(P, tag + "_pmc_hbm_traffic.csv"), "w").write("\n".join(lines) + "\n")
    # counter factors calibrated on the copy kernel of known size where the run has one (round 4: bench.py's same-box legs run
    # xHipMemCeilingDev's copy over the 2 GiB headline buffer), else the guide's gfx950 factor 2 for FETCH_SIZE
    cal = "mem_ceiling_kernel<copy>"
    f_fetch = (2048.0 * (1 << 20)) / (out[(cal, "FETCH_SIZE")] * 1024) if (cal, "FETCH_SIZE") in out else 2.0
    f_write = (2048.0 * (1 << 20)) / (out[(cal, "WRITE_SIZE")] * 1024) if (cal, "WRITE_SIZE") in out else 1.0
    T = lambda k: (f_fetch * out[(k, "FETCH_SIZE")] + f_write * out[(k, "WRITE_SIZE")]) * 1024
    traffic = {
        "_method": "rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE (TCC slots), KB per dispatch; gfx950 correction per "
                   "MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane streaming reads -> x2 (confirmed: the "
                   "forward DCT reads exactly 2 GiB and FETCH_SIZE reports 1.0000 GiB); WRITE_SIZE needs no correction (the fill kernel "
                   "writes exactly 2 GiB and reports 2097152.0 KB)",
        "_source": "profiles/%s_pmc_hbm_traffic.csv" % tag,
        "_factors": {"FETCH_SIZE": f_fetch, "WRITE_SIZE": f_write, "calibrated_on_copy_kernel": (cal, "FETCH_SIZE") in out},
        "dct32_fwd_bytes_per_launch": T("dct32_kernel<forward>"), "dct32_fwd_algorithmic_bytes": 4096 * (1 << 20),
        "dct32_inv_bytes_per_launch": T("dct32_kernel<inverse>"),
        "satd8x8_bytes_per_launch": T("satd8x8_kernel"), "satd8x8_algorithmic_bytes": 132 * (1 << 24)}
    json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    shutil.copy(bench, os.path.join(P, tag + "_bench.json"))
    if len(sys.argv) > 6:
        shutil.copy(sys.argv[6], os.path.join(P, tag + "_pytest_gpu.txt"))
    if len(sys.argv) > 8:
        sq_summary(tag, sys.argv[7], sys.argv[8])
    print(open(os.path.join(P, tag + "_pmc_hbm_traffic.csv")).read())
    for r in rows[:4]:
        print(r["Name"][:70], r["Calls"], "avg_ns", r["AverageNs"])


if __name__ == "__main__":
    main()
