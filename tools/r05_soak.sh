#!/bin/bash
# round-5 soak on the current build: random geometries on the default launch plan, then with the blur of every level / of levels 3.. forced into
# FAST's launch at these (small) batch sizes -- the plan 64-pair batches run by default -- and other patch sizes.  Progress lines keep gpurun's watchdog fed.
mkdir -p gpurun_out/r05
{
echo "build: $(python3 -c "import ctypes; l = ctypes.CDLL('orbslam2_amd/liborbfe.so'); l.orbfe_build_id.restype = ctypes.c_char_p; print(l.orbfe_build_id().decode())")"
SOAK_GEOM=1 SOAK_SEED=${SOAK_BASE:-710000} timeout -k 10 1000 python3 tools/soak.py ${1:-700}
ORBFE_BLUR_RIDE_FROM=0 SOAK_GEOM=1 SOAK_SEED=$((${SOAK_BASE:-710000} + 10000)) timeout -k 10 900 python3 tools/soak.py ${2:-500}
ORBFE_BLUR_RIDE_FROM=3 SOAK_PATCH=1 SOAK_SEED=$((${SOAK_BASE:-710000} + 20000)) timeout -k 10 600 python3 tools/soak.py ${3:-300}
} 2>&1 | grep --line-buffered -v "amdgpu.ids" | tee gpurun_out/r05/soak.txt
