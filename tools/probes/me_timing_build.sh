#!/bin/bash
# Developer build: libx266hip.so with per-workgroup cycle stamps in the motion search (me_search.hip, X266_ME_TIMING)
#   tools/probes/me_timing_build.sh   ->  tools/_ab/libx266hip_timing.so   (read by tools/probes/gpu_me_timing.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/x266_amd/csrc
make -C "$C" --no-print-directory >/dev/null
mkdir -p "$R/tools/_ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DX266_ME_TIMING -c -o "$R/tools/_ab/me_search_timing.o" "$C/me_search.hip"
OBJS=$(ls "$C"/build/*.o | grep -v me_search.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$R/tools/_ab/libx266hip_timing.so" $OBJS "$R/tools/_ab/me_search_timing.o" -Wl,--version-script="$C/libx266hip.map" -ldl
echo "built tools/_ab/libx266hip_timing.so"
