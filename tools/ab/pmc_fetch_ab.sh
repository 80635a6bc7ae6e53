# FETCH_SIZE / WRITE_SIZE per kernel for several builds (tools/ab/<name>.so); usage: pmc_fetch_ab.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp tools/ab/$v.so orbslam2_amd/liborbfe.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pf_$v_$c
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pf_${v}_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2>&1 || { echo "pass $v $c failed"; continue; }
    python3 tools/pmc_summary.py gpurun_out/pf_${v}_$c | grep -E "stereo_match|describe_k|fast_cell|octree3" | sed "s/^/$v /"
    rm -rf gpurun_out/pf_${v}_$c
  done
done
