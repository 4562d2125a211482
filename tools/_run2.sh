cd /root/repo
mkdir -p gpurun_out/r4b
timeout 900 tools/membench_read_shapes > gpurun_out/r4b/read_shapes.txt 2>&1
timeout 1500 python tools/gpu_satd_ceiling2.py > gpurun_out/r4b/satd_ceiling2.txt 2>&1
tail -3 gpurun_out/r4b/read_shapes.txt; tail -3 gpurun_out/r4b/satd_ceiling2.txt
