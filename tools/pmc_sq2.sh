#!/bin/bash
# SQ busy / stall counters per kernel (two passes of <= 8 counters), summaries under gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc2_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc2_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || { echo "pass $tag failed"; continue; }
  python3 tools/pmc_summary.py gpurun_out/pmc2_$tag > gpurun_out/pmc2_$tag.txt
  rm -rf gpurun_out/pmc2_$tag
done
cat gpurun_out/pmc2_*.txt | grep -v rocclr
