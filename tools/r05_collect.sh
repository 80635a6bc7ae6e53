#!/bin/bash
# round-5 collection on the GPU box: everything profiles/r05_* is made from -> gpurun_out/final/.  Order matters: the counter passes
# come first and are copied into profiles/ ON THE BOX, so that the bench.py run at the end replays them (their build id is the id
# of the library it runs: roofline.traffic / valu_issue are then present and labelled "replayed", not null + stale).
# Needs tools/ab/cuts.so (make -C orbslam2_amd/csrc cuts) for the per-phase instruction counts behind the opcode mix.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
B="--cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check"
bid=$(python3 -c "import ctypes; l = ctypes.CDLL('orbslam2_amd/liborbfe.so'); l.orbfe_build_id.restype = ctypes.c_char_p; print(l.orbfe_build_id().decode())") || exit 1
echo "build_id: $bid" | tee $out/build_id.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1 || { echo "pmc $c failed"; exit 1; }
  python3 tools/pmc_summary.py $out/pmc_$c $bid > $out/pmc_$c.txt; rm -rf $out/pmc_$c
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1 || { echo "pmc sq failed"; exit 1; }
python3 tools/pmc_summary.py $out/pmc_sq $bid > $out/pmc_sq.txt; rm -rf $out/pmc_sq
python3 tools/make_traffic.py $out 64 $out/traffic.json $bid > /dev/null
echo "counters done"
# per-phase instruction counts of FAST / describe on the cut-point build, then the opcode mix of THIS build's disassembly
if [ -f tools/ab/cuts.so ]; then
  cp orbslam2_amd/liborbfe.so $out/product.so; cp tools/ab/cuts.so orbslam2_amd/liborbfe.so
  bash tools/fast_insts.sh > $out/fast_insts.txt 2>/dev/null; bash tools/desc_insts.sh > $out/desc_insts.txt 2>/dev/null
  cp $out/product.so orbslam2_amd/liborbfe.so; rm $out/product.so
  cp $out/fast_insts.txt profiles/r05_fast_insts.txt; cp $out/desc_insts.txt profiles/r05_desc_insts.txt
  python3 tools/isa_mix.py $bid > $out/isa_mix.json 2>/dev/null && cp $out/isa_mix.json profiles/r05_isa_mix.json
  echo "isa mix done"
fi
for f in pmc_FETCH_SIZE.txt pmc_WRITE_SIZE.txt pmc_sq.txt traffic.json; do cp $out/$f profiles/r05_$f; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $B > $out/bench_under_rocprof.json 2>/dev/null || { echo "stats failed"; exit 1; }
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/stats
bash tools/step_trace.sh r05 > $out/step_trace.txt 2>/dev/null
bash tools/pmc_mem.sh $out/pmc_mem > /dev/null 2>&1; for f in $out/pmc_mem/TA_TA_BUSY_sum.txt $out/pmc_mem/TCP_PENDING_STALL_CYCLES_sum.txt; do [ -f $f ] && sed -i "1i # build_id: $bid" $f; done
echo "traces done"
python3 bench.py > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
timeout -k 10 300 python3 tools/pcie_rate2.py > $out/pcie.json 2>/dev/null
ORBFE_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --cpu-pairs 0 --host-fed 0 --natural 0 --small-batch 0 --secondary 0 --pipelined 0 > $out/bench_2rank_gloo_one_gpu.json 2> $out/bench_2rank.err; echo "2-rank self-launch rc=$?"
ORBFE_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 4 --backend gloo --cpu-pairs 0 --host-fed 0 --natural 0 --small-batch 0 --secondary 0 --pipelined 0 > $out/bench_4rank_gloo_one_gpu.json 2> $out/bench_4rank.err; echo "4-rank self-launch rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench.json"))
c = d["config"]; r = d["roofline"]
print("value %.0f ms %.4f n=%d pipelined %.0f small %.0f hostfed %.0f" % (d["value"], d["ms_per_step"], c["repeat"]["n"], c["pipelined"]["value"], c["small_batch"]["value"], c["host_fed"]["overlapped"]))
print("fast launch_ms", r["launch_ms"], "frac", r["frac"], "traffic", r.get("traffic"), "ratio", r.get("traffic_ratio"), "stale", r.get("stale_profiles"), "valu", (r.get("valu_issue") or {}).get("frac_mix"))
print(open("gpurun_out/final/step_trace.txt").read())
for n in (2, 4):
    try:
        s = json.load(open("gpurun_out/final/bench_%drank_gloo_one_gpu.json" % n))
        print(n, "ranks:", round(s["value"]), s["config"].get("strong_scaling"))
    except Exception as e:
        print(n, "ranks: no line", e)
PY
