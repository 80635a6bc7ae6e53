#!/bin/bash
# whole GPU suite on the default launch plan, then the frame-path tests on every alternative plan
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04/suite_default.log 2>&1; rc=$?; echo "default rc=$rc"; tail -3 gpurun_out/r04/suite_default.log
[ $rc -ne 0 ] && exit 1
F="tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_sweep.py tests/test_natural.py tests/test_golden.py tests/test_configs.py tests/test_round4_entry_points.py"
for plan in ORBFE_NO_PROC_ORDER=1 ORBFE_BLUR_IN_FAST=0 ORBFE_BK_DEPTH5=1 ORBFE_RS_LOOKUP=1 ORBFE_RS_LOOKUP=0 ORBFE_NO_PAIR=1 ORBFE_NO_PAIR=0 ORBFE_NO_INPLACE=1 ORBFE_PYR_LDS=1 ORBFE_NO_FUSE=1 ORBFE_NO_TAIL=1 ORBFE_NO_TAIL=0; do
  env $plan timeout -k 10 600 python -m pytest $F -m gpu -q -x > gpurun_out/r04/suite_$plan.log 2>&1; rc=$?; echo "$plan rc=$rc $(tail -1 gpurun_out/r04/suite_$plan.log)"
  [ $rc -ne 0 ] && { tail -30 gpurun_out/r04/suite_$plan.log; exit 1; }
done
exit 0
