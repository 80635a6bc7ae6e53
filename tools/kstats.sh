#!/bin/bash
# quick per-kernel timing on the GPU box: rocprofv3 kernel trace + stats of a short bench run -> gpurun_out/kstats.csv,
# and the pyramid launches split by level (launch order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kst; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kst -- python3 bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check --steps 10 > gpurun_out/kst.json 2>/dev/null || exit 1
cp gpurun_out/kst/*/*kernel_stats.csv gpurun_out/kstats.csv
python3 - <<'PY'
import csv, glob, collections, json
for r in csv.DictReader(open("gpurun_out/kstats.csv")):
    print(r["Name"][:34].ljust(34), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
f = glob.glob("gpurun_out/kst/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "pyr_resize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for i, r in enumerate(rows):
    acc[i % 7].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("pyr_resize by level:", [round(sum(v) / len(v), 1) for k, v in sorted(acc.items())])
allr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gaps = collections.defaultdict(list)
for a, b in zip(allr, allr[1:]):
    gaps[a["Kernel_Name"][:20] + " -> " + b["Kernel_Name"][:20]].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
print("gaps (us, median):")
for k, v in gaps.items():
    if len(v) >= 5: print("  ", k.ljust(46), round(sorted(v)[len(v) // 2], 1))
print(json.loads(open("gpurun_out/kst.json").read().strip().splitlines()[-1])["value"])
PY
rm -rf gpurun_out/kst
