cd /root/repo
mkdir -p gpurun_out/r4d
timeout 600 python -m pytest tests/test_gpu_mem_ceiling.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4d/pytest.txt
timeout 600 python tools/gpu_satd_pick.py > gpurun_out/r4d/satd_pick.txt 2>&1
timeout 600 python tools/gpu_host_probe2.py > gpurun_out/r4d/host_probe_reg0.txt 2>&1
X266HIP_HOST_REGISTER=1 timeout 600 python tools/gpu_host_probe2.py > gpurun_out/r4d/host_probe_reg1.txt 2>&1
timeout 900 tools/membench_stream_shapes > gpurun_out/r4d/stream_shapes.txt 2>&1
cat gpurun_out/r4d/pytest.txt; tail -3 gpurun_out/r4d/satd_pick.txt
