#!/bin/bash
# SQ counters of the motion-search kernels (run through gpurun): two PMC passes over tools/gpu_me_probe.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/me_pmc1_$T -- python $R/tools/gpu_me_probe.py ${2:-satd} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/me_pmc2_$T -- python $R/tools/gpu_me_probe.py ${2:-satd} > /dev/null 2>&1
cd $R
python - <<EOF
import csv, glob, collections, re
for d in ("gpurun_out/me_pmc1_$T", "gpurun_out/me_pmc2_$T"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "search" in r["Kernel_Name"]:
                m = re.search(r"(\w+search_kernel\w*<[^>]*>)", r["Kernel_Name"])
                k = (m.group(1) if m else r["Kernel_Name"][:50]) + " wg=%s vgpr=%s" % (r.get("Workgroup_Size", "?"), r.get("VGPR_Count", "?"))
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    tab = collections.defaultdict(dict)
    for (k, c), v in acc.items():
        tab[k][c] = sum(v) / len(v)
    for k, v in sorted(tab.items()):
        print(k)
        print("   ", "  ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
        if "GRBM_GUI_ACTIVE" in v:
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            print("    cycles %.4g  valu_busy %.3f  valu_instr_per_simd_cycle %.3f  wait_any_frac %.3f  wait_inst_frac %.3f" % (
                cyc, v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), v["SQ_INSTS_VALU"] / (1024 * cyc),
                v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
EOF
