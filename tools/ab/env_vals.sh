# same build, one environment variable at several values, 3 alternations; value + pipelined + per-stage table.
# usage: env_vals.sh VAR v1 v2 ...   ("-" = variable unset)
var=$1; shift
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 200 python bench.py --cpu-pairs 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/ab_ev_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ab_ev_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$var=$v', round(d['value']), 'piped', round(d['config']['pipelined']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
