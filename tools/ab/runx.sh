# general A/B: each variant = "name|ENV=val,ENV2=val|extra bench args" (name = tools/ab/<name>.so; env / args may be empty), 3 alternations
mkdir -p gpurun_out
for r in 1 2 3; do
  i=0
  for spec in "$@"; do
    i=$((i+1))
    IFS='|' read -r v e a <<< "$spec"
    cp tools/ab/$v.so orbslam2_amd/liborbfe.so
    ( for kv in $(echo $e | tr ',' ' '); do export $kv; done
      timeout -k 10 300 python bench.py --cpu-pairs 0 --host-fed 0 --small-batch 0 --natural 0 --secondary 0 --no-check $a > gpurun_out/abx_$i$r.json 2>/dev/null )
    python -c "
import json;d=json.loads(open('gpurun_out/abx_$i$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('%-36s' % '$spec', round(d['value']), 'piped', round(d['config']['pipelined']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
