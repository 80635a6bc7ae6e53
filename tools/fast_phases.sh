for d in 1 3 4 0; do
  ORBFE_FAST_DBG=$d timeout -k 10 120 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check --steps 10 > gpurun_out/fd$d.json 2> gpurun_out/fd$d.err
  python -c "
import json;d=json.loads(open('gpurun_out/fd$d.json').read().strip().splitlines()[-1]);print($d, d['roofline']['stage_ms_per_step_summed_over_groups']['fast'])"
done
