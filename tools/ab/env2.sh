# one build, several "VAR=val,VAR2=val2" settings ("-" = none), 3 alternations.  usage: env2.sh set1 set2 ...
mkdir -p gpurun_out
for r in 1 2 3; do
  i=0
  for st in "$@"; do
    i=$((i+1))
    ( if [ "$st" != "-" ]; then for kv in $(echo $st | tr ',' ' '); do export $kv; done; fi
      timeout -k 10 200 python bench.py --cpu-pairs 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/ab_e2_$i$r.json 2>/dev/null )
    python -c "
import json;d=json.loads(open('gpurun_out/ab_e2_$i$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('%-40s' % '$st', round(d['value']), 'piped', round(d['config']['pipelined']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
