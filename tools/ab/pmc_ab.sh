#!/bin/bash
# SQ counters of the FAST kernel for two builds (tools/ab/old.so, new.so) on one box: instruction counts and stall buckets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in old new; do
  cp tools/ab/$v.so orbslam2_amd/liborbfe.so
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf gpurun_out/pmcab_$v_$tag
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcab_${v}_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || exit 1
    echo "== $v $tag"; python3 tools/pmc_summary.py gpurun_out/pmcab_${v}_$tag | grep -E "fast_cell|describe_k"
    rm -rf gpurun_out/pmcab_${v}_$tag
  done
done
