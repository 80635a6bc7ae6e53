#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + HBM traffic counters, summaries into gpurun_out/final/.
# The profiled passes skip bench.py's host-fed extra (--host-fed 0): it runs two contexts at once, whose overlapping kernels would
# inflate the per-kernel averages that must agree with the timed region's launch_ms.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no sys/hip tracing with --pmc).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > $out/bench_under_rocprof.json 2>/dev/null || exit 1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/pmc_$c > $out/pmc_$c.txt
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pmc_sq > $out/pmc_sq.txt
python3 tools/make_traffic.py $out 64 $out/traffic.json > /dev/null
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq
cat $out/bench.json; cut -c1-120 $out/kernel_stats.csv; cat $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt | grep -v rocclr
