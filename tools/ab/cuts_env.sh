#!/bin/bash
# the cut-point build (tools/ab/cuts.so), one environment variable at several values.  usage: cuts_env.sh VAR v1 v2 ...
mkdir -p gpurun_out
cp orbslam2_amd/liborbfe.so /tmp/liborbfe.keep && cp tools/ab/cuts.so orbslam2_amd/liborbfe.so
var=$1; shift
for r in 1 2; do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/ce_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ce_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$var=$v', round(d['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
cp /tmp/liborbfe.keep orbslam2_amd/liborbfe.so
