#!/bin/bash
# whole GPU suite on the default launch plan, then the frame-path tests on every alternative plan that is left (round 5: six).
# ORBFE_OCTREE=1 forces the generic quadtree kernel, whose LDS budget refuses large per-level quotas: contexts it cannot create
# are skipped by the tests (ORBFE_ERR_UNSUPPORTED), never silently rerouted.
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r05/suite_default.log 2>&1; rc=$?; echo "default rc=$rc $(tail -1 gpurun_out/r05/suite_default.log)"
[ $rc -ne 0 ] && { tail -30 gpurun_out/r05/suite_default.log; exit 1; }
F="tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_sweep.py tests/test_natural.py tests/test_golden.py tests/test_configs.py tests/test_round4_entry_points.py tests/test_round5_entry_points.py"
for plan in ORBFE_NO_INPLACE=1 "ORBFE_NO_PAIR=1 ORBFE_NO_TAIL=1" ORBFE_PYR_LDS=1 ORBFE_NO_FUSE=1 ORBFE_NO_PROC_ORDER=1 ORBFE_OCTREE=1; do
  tag=$(echo $plan | tr ' ' '+')
  env $plan timeout -k 10 600 python -m pytest $F -m gpu -q -x > gpurun_out/r05/suite_$tag.log 2>&1; rc=$?; echo "$tag rc=$rc $(tail -1 gpurun_out/r05/suite_$tag.log)"
  [ $rc -ne 0 ] && { tail -40 gpurun_out/r05/suite_$tag.log; exit 1; }
done
exit 0
