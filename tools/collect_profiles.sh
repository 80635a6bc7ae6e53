#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + HBM traffic counters, summaries into gpurun_out/final/.
# The profiled passes skip bench.py's host-fed extra (--host-fed 0): it runs two contexts at once, whose overlapping kernels would
# inflate the per-kernel averages that must agree with the timed region's launch_ms.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no sys/hip tracing with --pmc).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
# every counter file carries the id of the build it was taken on (orbfe_build_id(): sha256 over the library's sources and flags);
# bench.py replays roofline.traffic / valu_issue only from files whose id equals the id of the library it has just run
bid=$(python3 -c "import ctypes; l = ctypes.CDLL('orbslam2_amd/liborbfe.so'); l.orbfe_build_id.restype = ctypes.c_char_p; print(l.orbfe_build_id().decode())") || exit 1
echo "build_id: $bid" > $out/build_id.txt
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > $out/bench_under_rocprof.json 2>/dev/null || exit 1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/pmc_$c $bid > $out/pmc_$c.txt
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check  > /dev/null 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pmc_sq $bid > $out/pmc_sq.txt
python3 tools/make_traffic.py $out 64 $out/traffic.json $bid > /dev/null
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq
cat $out/bench.json; cut -c1-120 $out/kernel_stats.csv; cat $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt | grep -v rocclr
