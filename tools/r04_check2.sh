#!/bin/bash
mkdir -p gpurun_out/r04
T="tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_batch.py tests/test_natural.py"
timeout -k 10 500 python -m pytest $T -m gpu -q -x > gpurun_out/r04/t_inplace.log 2>&1; rc=$?; echo "inplace rc=$rc"; tail -4 gpurun_out/r04/t_inplace.log
[ $rc -ne 0 ] && exit 1
bash tools/step_trace.sh copy ORBFE_NO_INPLACE=1 ORBFE_NO_FUSE=1 | grep -E " (0|1|8|7) |sum"
bash tools/step_trace.sh inplace ORBFE_NO_FUSE=1 | grep -E " (0|1|7) |sum"
bash tools/ab/env_ab.sh ORBFE_NO_INPLACE
