#!/bin/bash
# round-4 collection on the GPU box: everything profiles/r04_* is made from -> gpurun_out/final/
bash tools/collect_profiles.sh > /dev/null 2>&1 || { echo "collect_profiles failed"; tail -5 gpurun_out/final/bench.err; exit 1; }
out=gpurun_out/final
bash tools/step_trace.sh r04 > $out/step_trace.txt 2>/dev/null
timeout -k 10 300 python3 tools/pcie_rate2.py > $out/pcie.json 2>/dev/null
[ -f tools/ab/cuts.so ] && bash tools/octree3_timeline.sh 2>/dev/null | grep -v amdgpu.ids > $out/octree3_timeline.txt
ORBFE_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --cpu-pairs 0 --host-fed 0 --natural 0 --small-batch 0 --secondary 0 --pipelined 0 > $out/bench_2rank_gloo_one_gpu.json 2> $out/bench_2rank.err; echo "2-rank self-launch rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
c = d["config"]
print("value %.0f ms %.4f pipelined %.0f small %.0f hostfed %.0f (unpacked %.0f left %.0f)" % (d["value"], d["ms_per_step"], c["pipelined"]["value"], c["small_batch"]["value"],
      c["host_fed"]["overlapped"], c["host_fed"]["unpacked"]["overlapped"], c["host_fed"]["left_only"]["overlapped"]))
print("fast launch_ms", d["roofline"]["launch_ms"], "traffic_ratio", d["roofline"].get("traffic_ratio"), d["roofline"]["whole_pipeline"].get("traffic_ratio"))
print(open("gpurun_out/final/step_trace.txt").read())
print(open("gpurun_out/final/bench_2rank_gloo_one_gpu.json").read()[:400])
PY
