#!/bin/bash
# after `gpurun -- bash tools/r05_collect.sh`: gpurun_out/final/* -> profiles/r05_* (only what that script wrote)
f=gpurun_out/final
for n in bench.json bench_under_rocprof.json kernel_stats.csv step_trace.txt pmc_FETCH_SIZE.txt pmc_WRITE_SIZE.txt pmc_sq.txt traffic.json isa_mix.json fast_insts.txt desc_insts.txt pcie.json bench_2rank_gloo_one_gpu.json bench_4rank_gloo_one_gpu.json; do
  [ -f $f/$n ] && cp $f/$n profiles/r05_$n
done
[ -f $f/pmc_mem/TA_TA_BUSY_sum.txt ] && cp $f/pmc_mem/TA_TA_BUSY_sum.txt profiles/r05_pmc_mem_ta.txt
[ -f $f/pmc_mem/TCP_PENDING_STALL_CYCLES_sum.txt ] && cp $f/pmc_mem/TCP_PENDING_STALL_CYCLES_sum.txt profiles/r05_pmc_mem_tcp.txt
cat $f/build_id.txt; grep -l "$(cut -d' ' -f2 $f/build_id.txt)" profiles/r05_* | wc -l
