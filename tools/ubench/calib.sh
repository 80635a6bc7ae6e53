cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip || exit 1
rm -rf gpurun_out/cal; rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/cal -- /tmp/valu_rate > gpurun_out/cal.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/cal; cat gpurun_out/cal.txt | tail -8; rm -rf gpurun_out/cal
