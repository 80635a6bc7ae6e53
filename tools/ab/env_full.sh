# same build, one environment variable on / off, 3 alternations, with the pipelined and small-batch regimes.  usage: env_full.sh VAR
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in 1 0; do
    env $1=$v timeout -k 10 300 python bench.py --cpu-pairs 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/ab_envf$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ab_envf$v$r.json').read().strip().splitlines()[-1]);c=d['config'];print('$1=$v', round(d['value']), 'piped', round(c['pipelined']['value']), 'small', round(c['small_batch']['value']), 'small one chain', round(c['small_batch']['one_chain_at_a_time']))"
  done
done
