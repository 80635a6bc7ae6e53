#!/bin/bash
# tools/octree3_timeline.py on the cut-point build (`make -C orbslam2_amd/csrc cuts` first; the product library is put back afterwards)
mkdir -p gpurun_out
cp orbslam2_amd/liborbfe.so /tmp/liborbfe.keep && cp tools/ab/cuts.so orbslam2_amd/liborbfe.so
for a in "$@"; do export "$a"; done
timeout -k 10 300 python3 tools/octree3_timeline.py; rc=$?
cp /tmp/liborbfe.keep orbslam2_amd/liborbfe.so
exit $rc
