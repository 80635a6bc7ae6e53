#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_round4_entry_points.py tests/test_abi.py tests/test_configs.py -m gpu -q -x > gpurun_out/r04/t_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -15 gpurun_out/r04/t_new.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python bench.py > gpurun_out/r04/bench_full.json 2> gpurun_out/r04/bench_full.err; echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_full.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/bench_full.json").read().strip().splitlines()[-1])
c = d["config"]
print("value %.0f ms %.4f pipelined %.0f small %.0f" % (d["value"], d["ms_per_step"], c["pipelined"]["value"], c["small_batch"]["value"]))
print("host_fed", json.dumps(c.get("host_fed"))[:1500])
print("secondary", json.dumps(c.get("secondary"))[:2500])
print("single_frame", json.dumps(c.get("single_frame")))
print("rccl", c.get("rccl"), "traffic_ratio", d["roofline"].get("traffic_ratio"), d["roofline"]["whole_pipeline"].get("traffic_ratio"))
PY
