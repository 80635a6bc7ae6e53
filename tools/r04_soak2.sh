#!/bin/bash
# second soak of round 4: new seeds, plus the depth-5 quadtree plan and forced level pairs / table look-ups
mkdir -p gpurun_out/r04
{
echo "build: $(sha256sum orbslam2_amd/liborbfe.so | cut -c1-16)"
SOAK_GEOM=1 SOAK_SEED=510000 timeout -k 10 900 python3 tools/soak.py ${1:-800}
SOAK_SEED=520000 timeout -k 10 600 python3 tools/soak.py ${2:-400}
ORBFE_BK_DEPTH5=1 SOAK_GEOM=1 SOAK_SEED=530000 timeout -k 10 600 python3 tools/soak.py ${3:-200}
ORBFE_NO_PAIR=1 ORBFE_RS_LOOKUP=1 SOAK_GEOM=1 SOAK_SEED=540000 timeout -k 10 600 python3 tools/soak.py ${4:-200}
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r04/soak2.txt
