# average duration of the kernels matching a pattern for one build: usage ktime.sh <name> <grep pattern>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp tools/ab/$1.so orbslam2_amd/liborbfe.so
[ -n "$3" ] && export $3
rm -rf gpurun_out/kt_$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$1 -- python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2>&1 || exit 1
f=$(find gpurun_out/kt_$1 -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv,re,sys
for r in csv.DictReader(open('$f')):
    if re.search('$2', r['Name']): print('$1 $3', r['Name'].split('(')[0][-32:], r['Calls'], round(float(r['AverageNs'])/1000,1), 'us')"
rm -rf gpurun_out/kt_$1
