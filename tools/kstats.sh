#!/bin/bash
# quick per-kernel timing on the GPU box: rocprofv3 kernel stats of a short bench run -> gpurun_out/kstats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kst; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kst -- python3 bench.py --cpu-pairs 0 --no-check --steps 10 > gpurun_out/kst.json 2>/dev/null || exit 1
cp gpurun_out/kst/*/*kernel_stats.csv gpurun_out/kstats.csv; rm -rf gpurun_out/kst
cut -d, -f1-4 gpurun_out/kstats.csv | cut -c1-110
