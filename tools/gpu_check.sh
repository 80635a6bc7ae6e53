#!/bin/bash
# GPU-box helper: parity subset + bench summary.  usage: tools/gpu_check.sh <tag> [pytest paths...]
tag=$1; shift
mkdir -p gpurun_out/r03
tests=${@:-tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_natural.py tests/test_gpu_batch.py tests/test_golden.py}
python -m pytest $tests -m gpu -q -x > gpurun_out/r03/t_$tag.log 2>&1; tail -6 gpurun_out/r03/t_$tag.log
python bench.py --cpu-pairs 0 --host-fed 0 > gpurun_out/r03/bench_$tag.json 2> gpurun_out/r03/bench_$tag.err || { tail -5 gpurun_out/r03/bench_$tag.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_$tag.json"))
print("value %.0f  pipelined %.0f  small_batch %.0f  fast launch_ms %.4f" % (d["value"], d["config"]["pipelined"]["value"], d["config"]["small_batch"]["value"], d["roofline"]["launch_ms"]))
print({k: round(v, 4) for k, v in d["roofline"]["stage_ms_per_step_summed_over_groups"].items()})
PY
