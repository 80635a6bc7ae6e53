# same build, one environment variable on / off, 3 alternations; headline stages + the photograph's rate (with the oracle check).  usage: env_nat.sh VAR
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in 1 0; do
    env $1=$v timeout -k 10 300 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --host-fed 0 --secondary 0 > gpurun_out/ab_envn$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ab_envn$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$1=$v', round(d['value']), 'natural', round(d['config']['natural_image']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
