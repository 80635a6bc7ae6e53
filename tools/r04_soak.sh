#!/bin/bash
# round-4 soak on the current build: random geometries (default launch plan = in place, level pairs, 256-thread quadtree), then the copy plan
mkdir -p gpurun_out/r04
{
echo "build: $(sha256sum orbslam2_amd/liborbfe.so | cut -c1-16)"
SOAK_GEOM=1 SOAK_SEED=410000 timeout -k 10 900 python3 tools/soak.py ${1:-700}
SOAK_SEED=420000 timeout -k 10 600 python3 tools/soak.py ${2:-400}
SOAK_PATCH=1 SOAK_GEOM=1 SOAK_SEED=430000 timeout -k 10 600 python3 tools/soak.py ${3:-300}
ORBFE_NO_INPLACE=1 ORBFE_NO_PAIR=1 SOAK_GEOM=1 SOAK_SEED=440000 timeout -k 10 600 python3 tools/soak.py ${4:-200}
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/soak.txt
