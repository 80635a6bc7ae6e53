# A/B/... on one box: alternate several builds of liborbfe.so (tools/ab/<name>.so), 3 rounds each.  usage: runv.sh name1 name2 ...
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in "$@"; do
    cp tools/ab/$v.so orbslam2_amd/liborbfe.so
    timeout -k 10 300 python bench.py --cpu-pairs 0 --host-fed 0 --secondary 0 --no-check  > gpurun_out/ab_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ab_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$v', round(d['value']), 'piped', round(d['config']['pipelined']['value']), 'small', round(d['config']['small_batch']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
