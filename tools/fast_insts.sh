#!/bin/bash
# VALU / LDS / SALU instructions per wave (= per FAST cell) of fast_cell_kernel at its debug cut points
# (1: tile fill, 3: + phase A, 4: + phase C, 0: whole kernel) -- instruction counts, not times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 1 3 4 0; do
  rm -rf gpurun_out/fi$d
  ORBFE_FAST_DBG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/fi$d -- python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/fi$d | grep fast_cell | python3 -c "
import sys,ast
l=sys.stdin.read(); d=ast.literal_eval(l[l.index('{'):l.rindex('}')+1]); w=d['SQ_WAVES']
print('dbg $d per wave:', {k: round(v/w,1) for k,v in d.items() if k!='SQ_WAVES'})"
  rm -rf gpurun_out/fi$d
done
