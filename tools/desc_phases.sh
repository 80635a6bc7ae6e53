# cumulative time of describe_kernel's phases (debug early-exits; outputs are invalid in these runs)
for d in 1 2 3 0; do
  ORBFE_DESC_DBG=$d timeout -k 10 120 python bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check --steps 10 > gpurun_out/dd$d.json 2> gpurun_out/dd$d.err
  python -c "
import json;d=json.loads(open('gpurun_out/dd$d.json').read().strip().splitlines()[-1]);print($d, d['roofline']['stage_ms_per_step_summed_over_groups']['describe'])"
done
