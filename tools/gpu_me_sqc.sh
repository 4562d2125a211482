#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# Scalar data-cache and instruction-cache counters of the motion-search kernels (run through gpurun): one PMC pass over tools/gpu_me_probe.py
cd /tmp && export TMPDIR=/tmp

rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d $R/gpurun_out/me_sqc -- python $R/tools/gpu_me_probe.py satd > /dev/null 2>&1
cd $R
python - <<EOF
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/me_sqc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "search" in r["Kernel_Name"]:
            m = re.search(r"(\w+search_kernel\w*<[^>]*>)", r["Kernel_Name"])
            acc[(m.group(1) if m else r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "%.4g" % (sum(v) / len(v)))
EOF
