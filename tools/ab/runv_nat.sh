# A/B/... on one box incl. the photograph: alternate builds of liborbfe.so (tools/ab/<name>.so), 3 rounds each, oracle check on.  usage: runv_nat.sh name1 name2 ...
mkdir -p gpurun_out
cp orbslam2_amd/liborbfe.so /tmp/liborbfe.keep
for r in 1 2 3; do
  for v in "$@"; do
    cp tools/ab/$v.so orbslam2_amd/liborbfe.so
    timeout -k 10 300 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --host-fed 0 --secondary 0 > gpurun_out/abn_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/abn_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$v', round(d['value']), 'natural', round(d['config']['natural_image']['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
cp /tmp/liborbfe.keep orbslam2_amd/liborbfe.so
