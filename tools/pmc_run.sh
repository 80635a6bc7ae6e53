#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <counters...>   (run on the GPU box; counters in their own pass)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 3 --warmup 1 --pairs 32 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > gpurun_out/$out.json 2> gpurun_out/$out.err
python3 tools/pmc_summary.py gpurun_out/$out
