#!/bin/bash
mkdir -p gpurun_out/r04
T="tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_batch.py tests/test_natural.py tests/test_golden.py tests/test_round4_entry_points.py"
for e in ORBFE_RS_LOOKUP=1 ORBFE_RS_LOOKUP=0; do
env $e timeout -k 10 600 python -m pytest $T -m gpu -q -x > gpurun_out/r04/t_$e.log 2>&1; rc=$?; echo "$e rc=$rc $(tail -1 gpurun_out/r04/t_$e.log)"
[ $rc -ne 0 ] && { tail -20 gpurun_out/r04/t_$e.log; exit 1; }
done
cp orbslam2_amd/liborbfe.so /tmp/keep.so
bash tools/ab/runv.sh old new
cp /tmp/keep.so orbslam2_amd/liborbfe.so
bash tools/step_trace.sh cur | grep -E "pyr|sum"
