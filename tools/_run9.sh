cd /root/repo
mkdir -p gpurun_out/r4i
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4i/pytest_gpu.txt
timeout 600 python -m pytest tests/test_gpu_perf_floor.py -q -m gpu -s 2>&1 | grep -E "box|fraction|search|passed|failed" > gpurun_out/r4i/perf_floor.txt
timeout 900 python bench.py > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
timeout 600 python tools/gpu_satd_batch.py sizes > gpurun_out/r4i/satd_batch_sizes.txt 2>&1
cat gpurun_out/r4i/pytest_gpu.txt; cat gpurun_out/r4i/perf_floor.txt; tail -3 gpurun_out/r4i/bench.err; head -c 600 gpurun_out/r4i/bench.json
