# A/B on one box: alternate two builds of liborbfe.so (tools/ab/old.so, new.so), 3 rounds each
for r in 1 2 3; do
  for v in old new; do
    cp tools/ab/$v.so orbslam2_amd/liborbfe.so
    timeout -k 10 200 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/ab_$v$r.json 2>/dev/null
    python -c "
import json;d=json.loads(open('gpurun_out/ab_$v$r.json').read().strip().splitlines()[-1]);s=d['roofline']['stage_ms_per_step_summed_over_groups'];print('$v', round(d['value']), ' '.join('%s=%.4f' % (k[:4], x) for k, x in s.items()))"
  done
done
