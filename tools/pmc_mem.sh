#!/bin/bash
# memory-pipe counters (texture addresser / L1 / L2 request counts) per kernel, one pass per counter group.
# usage: tools/pmc_mem.sh <outdir>   (run on the GPU box)
out=${1:-gpurun_out/pmc_mem}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+\b" | sort -u > $out/avail.txt
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum" \
           ; do  # a third set (TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum ...) aborted rocprofv3 and hung its shutdown: check names against avail.txt first
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $out/raw_$tag
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/raw_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2> $out/err_$tag.txt || { echo "pass $tag failed"; tail -3 $out/err_$tag.txt; continue; }
  python3 tools/pmc_summary.py $out/raw_$tag > $out/$tag.txt
  rm -rf $out/raw_$tag
done
cat $out/*_sum.txt $out/TA_*.txt 2>/dev/null | grep -E "describe_k|fast_cell|pyr_resize_direct|blur_k|stereo_match" | sort | uniq
