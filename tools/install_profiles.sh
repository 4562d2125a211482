#!/bin/bash
# After tools/run_profiles.sh <round> <suffix> has come back through gpurun: copy its outputs into profiles/ and regenerate profiles/MEASURED.md and DESIGN.md section 0.
#   tools/install_profiles.sh <round-tag, e.g. r04> <suffix>
set -e
T=${1:?round tag}; S=${2:?suffix}
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python tools/collect_profiles.py $T gpurun_out/prof_$S gpurun_out/pmc_fetch_$S gpurun_out/pmc_write_$S gpurun_out/bench_${T}_$S.json gpurun_out/pytest_gpu.log gpurun_out/pmc_sq_$S gpurun_out/pmc_lds_$S | tail -1
cp gpurun_out/bench_${T}_$S.json profiles/${T}_bench.json
cp gpurun_out/bench_${T}_${S}_driver_args.json profiles/${T}_bench_driver_args.json
cp gpurun_out/stream8k_${T}_$S.json profiles/${T}_stream8k.json
cp gpurun_out/bench_${T}_${S}_line.json profiles/${T}_bench_line.json                      # the compact line, as the driver sees it
cp gpurun_out/bench_${T}_${S}_driver_args_line.json profiles/${T}_bench_driver_args_line.json
cp gpurun_out/bench_${T}_${S}_profiled_run.json profiles/${T}_bench_profiled_run.json
python tools/design_table.py profiles/${T}_bench.json --write
python - "$T" <<'PY'
import csv, json, re, sys
t = sys.argv[1]
d = json.loads(open("profiles/%s_bench_profiled_run.json" % t).read())
rows = list(csv.reader(open("profiles/%s_bench_kernel_stats.csv" % t)))
f = [r for r in rows if "dct32_lds_kernel<false>" in r[0]][0]
s = [r for r in rows if "satd8x8_dma" in r[0]][0]
new = "within 1–2 %%: forward %.4f ms over %s launches against %.4f, SATD %.4f over %s against\n%.4f" % (
    float(f[3]) / 1e6, f[1], d["roofline"]["kernel_ms_per_launch"], float(s[3]) / 1e6, s[1], d["also"]["satd8x8"]["kernel_ms"])
D = open("profiles/MEASURED.md").read()
m = re.search(r"within 1–2 %: forward [0-9.]+ ms over \d+\s+launches against [0-9.]+, SATD [0-9.]+ over \d+\s+against\s+[0-9.]+", D)
open("profiles/MEASURED.md", "w").write(D.replace(m.group(0), new))
print(new)
PY
