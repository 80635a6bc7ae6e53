#!/bin/bash
# FAST wide-LDS experiment: parity of the variant, A/B against the shipped build, LDS instruction counters of both
mkdir -p gpurun_out/r04
cp orbslam2_amd/liborbfe.so /tmp/keep.so
cp tools/ab/wide.so orbslam2_amd/liborbfe.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_natural.py -m gpu -q -x > gpurun_out/r04/t_wide.log 2>&1; rc=$?; echo "wide parity rc=$rc"; tail -3 gpurun_out/r04/t_wide.log
[ $rc -ne 0 ] && { cp /tmp/keep.so orbslam2_amd/liborbfe.so; exit 1; }
bash tools/ab/runv.sh base wide
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base wide; do
  cp tools/ab/$v.so orbslam2_amd/liborbfe.so
  rm -rf gpurun_out/pmcx
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmcx -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2>&1
  echo "$v: $(python3 tools/pmc_summary.py gpurun_out/pmcx | grep fast_cell)"
done
rm -rf gpurun_out/pmcx
cp /tmp/keep.so orbslam2_amd/liborbfe.so
