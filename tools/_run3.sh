cd /root/repo
mkdir -p gpurun_out/r4c
timeout 2400 python tools/gpu_satd_ceiling3.py > gpurun_out/r4c/satd_ceiling3.txt 2>&1
tail -3 gpurun_out/r4c/satd_ceiling3.txt
