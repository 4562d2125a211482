#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun; outputs under gpurun_out/).
#   tools/run_profiles.sh <round-tag, e.g. r02> <suffix>
# Kernel-trace + stats in one pass; every PMC group in its own pass (never with sys/hip/hsa traces); every pass under its own
# timeout (a hung counter pass once cost a whole gpurun budget slice).
T=${1:-r04}
S=${2:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-live-traffic --no-host-api --no-autotune"   # (bench.py also refuses to nest a profiler on its own)
# --stream8k 0: the node legs launch the same kernels on frame-sized batches, which would mix sizes into the per-kernel averages
# bench.py prints the compact line the driver parses and leaves the full record in ./bench_full.json (here: /tmp)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$S -- $B --stream8k 0 > $R/gpurun_out/prof_$S.log 2>&1
cp /tmp/bench_full.json $R/gpurun_out/bench_${T}_${S}_profiled_run.json
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$S -- $B --no-me --no-transform-set --stream8k 0 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$S -- $B --no-me --no-transform-set --stream8k 0 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_$S -- $B --stream8k 0 --steps 5 --warmup 2 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_lds_$S -- $B --stream8k 0 --steps 5 --warmup 2 > /dev/null 2>&1
cd $R
python bench.py > gpurun_out/bench_${T}_${S}_line.json; cp bench_full.json gpurun_out/bench_${T}_$S.json
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${T}_${S}_driver_args_line.json; cp bench_full.json gpurun_out/bench_${T}_${S}_driver_args.json
wc -c gpurun_out/bench_${T}_${S}_line.json gpurun_out/bench_${T}_${S}_driver_args_line.json
python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -6 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
./host/stream8k 0 500 | tail -1 > gpurun_out/stream8k_${T}_$S.json 2>/dev/null; cat gpurun_out/stream8k_${T}_$S.json
