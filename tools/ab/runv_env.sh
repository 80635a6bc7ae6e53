# like runv.sh, with an environment assignment per variant: usage runv_env.sh name[:VAR=val] ...
mkdir -p gpurun_out
for r in 1 2 3; do
  for spec in "$@"; do
    v=${spec%%:*}; e=""; [ "$spec" != "$v" ] && e=${spec#*:}
    cp tools/ab/$v.so orbslam2_amd/liborbfe.so
    env $e timeout -k 10 300 python bench.py --cpu-pairs 0 --host-fed 0 --small-batch 0 --natural 0 --no-check > gpurun_out/abe_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/abe_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$spec', round(d['value']), 'piped', round(d['config']['pipelined']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()), 'p+b=%.4f' % (s['pyramid'] + s['blur']))"
  done
done
