set -x
cd /root/repo
mkdir -p gpurun_out/r4a
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "satd" 2>&1 | tail -5 > gpurun_out/r4a/pytest_satd.txt
timeout 1500 python tools/gpu_satd_ceiling.py > gpurun_out/r4a/satd_ceiling.txt 2>&1
tail -5 gpurun_out/r4a/pytest_satd.txt
head -40 gpurun_out/r4a/satd_ceiling.txt
