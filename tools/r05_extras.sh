#!/bin/bash
# the per-call / per-frame measurements of the rows outside the benchmarked step, on the final build
mkdir -p gpurun_out/r05x
timeout -k 10 300 python3 tools/latency.py > gpurun_out/r05x/latency.json 2> gpurun_out/r05x/latency.err; echo "latency rc=$?"
timeout -k 10 300 python3 tools/bench_matchers.py > gpurun_out/r05x/matchers.json 2> gpurun_out/r05x/matchers.err; echo "matchers rc=$?"
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r05x/configs.json 2> gpurun_out/r05x/configs.err; echo "configs rc=$?"
timeout -k 10 300 python3 tools/small_batch.py --pairs 8 --chains 1,2,3,4,6 > gpurun_out/r05x/small_batch.json 2> gpurun_out/r05x/sb.err; echo "small_batch rc=$?"
timeout -k 10 300 python3 tools/soak_pose.py 300 > gpurun_out/r05x/soak_pose.log 2>&1; echo "soak_pose rc=$?"; tail -2 gpurun_out/r05x/soak_pose.log
head -12 gpurun_out/r05x/latency.json
