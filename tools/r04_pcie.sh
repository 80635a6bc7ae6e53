#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 500 python3 tools/pcie_rate2.py > gpurun_out/r04/pcie2.json 2> gpurun_out/r04/pcie2.err; echo rc=$?; tail -3 gpurun_out/r04/pcie2.err
python3 -c "
import json
d=json.load(open('gpurun_out/r04/pcie2.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v['pairs_per_s_median'], v['passes'], v.get('bytes_down_per_step'), v.get('equals_unpacked'))
"
