#!/bin/bash
mkdir -p gpurun_out/r04
s=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_timed.json 2> gpurun_out/r04/bench_timed.err; echo "rc=$?"
e=$(date +%s.%N)
echo "wall $(echo "$e - $s" | bc) s"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_timed.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['host_fed']['overlapped'], d['config']['pipelined']['value'], d['config']['secondary'].get('error'))"
