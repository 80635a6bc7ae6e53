#!/bin/bash
mkdir -p gpurun_out/r04
{
echo "build: $(sha256sum orbslam2_amd/liborbfe.so | cut -c1-16)"
SOAK_GEOM=1 SOAK_SEED=${SOAK_BASE:-610000} timeout -k 10 1000 python3 tools/soak.py ${1:-1500}
SOAK_PATCH=1 SOAK_SEED=$((${SOAK_BASE:-610000} + 10000)) timeout -k 10 600 python3 tools/soak.py ${2:-500}
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/soak3.txt
