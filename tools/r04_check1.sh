#!/bin/bash
# round 4, first GPU check: frame-path parity with level 0 read in place (default) and with the ingest copy, then A/B of the two
mkdir -p gpurun_out/r04
T="tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_natural.py tests/test_gpu_batch.py tests/test_golden.py tests/test_configs.py"
timeout -k 10 500 python -m pytest $T -m gpu -q -x > gpurun_out/r04/t_inplace.log 2>&1; echo "inplace rc=$?"; tail -4 gpurun_out/r04/t_inplace.log
ORBFE_NO_INPLACE=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x > gpurun_out/r04/t_copy.log 2>&1; echo "copy rc=$?"; tail -3 gpurun_out/r04/t_copy.log
bash tools/ab/env_ab.sh ORBFE_NO_INPLACE
bash tools/kstats.sh
