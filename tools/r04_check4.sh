#!/bin/bash
# parity of the current build on the frame path, then A/B old.so vs new.so (headline + pipelined + small batch)
mkdir -p gpurun_out/r04
T="tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_batch.py tests/test_natural.py tests/test_golden.py tests/test_configs.py tests/test_round4_entry_points.py"
timeout -k 10 600 python -m pytest $T -m gpu -q -x > gpurun_out/r04/t_cur.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r04/t_cur.log
[ $rc -ne 0 ] && exit 1
cp orbslam2_amd/liborbfe.so /tmp/keep.so
bash tools/ab/runv.sh old new
cp /tmp/keep.so orbslam2_amd/liborbfe.so
bash tools/step_trace.sh cur | grep -E "octree|sum"
