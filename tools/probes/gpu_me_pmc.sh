#!/bin/bash
# SQ / SQC counters of the motion-search kernels (run through gpurun): three PMC passes over tools/probes/gpu_me_probe.py
# usage: gpu_me_pmc.sh <tag> [satd|sad|all]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-x}
W=${2:-satd}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/me_pmc1_$T -- python $R/tools/probes/gpu_me_probe.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/me_pmc2_$T -- python $R/tools/probes/gpu_me_probe.py $W > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d $R/gpurun_out/me_pmc3_$T -- python $R/tools/probes/gpu_me_probe.py $W > /dev/null 2>&1
cd $R
python - <<EOF
import csv, glob, collections, re
tab = collections.defaultdict(dict)
for d in ("gpurun_out/me_pmc1_$T", "gpurun_out/me_pmc2_$T", "gpurun_out/me_pmc3_$T"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "search" in r["Kernel_Name"]:
                m = re.search(r"(\w+search_kernel\w*<[^>]*>)", r["Kernel_Name"])
                k = (m.group(1) if m else r["Kernel_Name"][:50]) + " wg=%s vgpr=%s" % (r.get("Workgroup_Size", "?"), r.get("VGPR_Count", "?"))
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
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
    if "SQC_DCACHE_REQ" in v and v["SQC_DCACHE_REQ"]:
        print("    scalar cache: miss rate %.3f" % (v["SQC_DCACHE_MISSES"] / v["SQC_DCACHE_REQ"]))
EOF
