#!/bin/bash
# per-launch durations of one step, in launch order (median over the steps of a short bench run under rocprofv3 --kernel-trace).
# usage: step_trace.sh <tag> [ENV=VAL ...]   (the environment assignments apply to the bench run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for a in "$@"; do export "$a"; done
rm -rf gpurun_out/st_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st_$tag -- python3 bench.py --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check --steps 10 --repeat 2 > gpurun_out/st_$tag.json 2>/dev/null || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("gpurun_out/st_%s/*/*kernel_trace.csv" % tag)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rocclr" not in r["Kernel_Name"] and "candidates_gather" not in r["Kernel_Name"]]
steps, cur = [], []
last = "stereo_median" if any("stereo_median" in r["Kernel_Name"] for r in rows) else "stereo_match"  # the step's last launch (round 5: the median cut rides in the matcher's launch)
for r in rows:
    cur.append(r)
    if last in r["Kernel_Name"]:
        steps.append(cur); cur = []
n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == n][2:]
tot = 0.0
for i in range(n):
    d = sorted((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in steps)
    g = sorted((int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"])) / 1e3 for s in steps) if i else [0.0]
    tot += d[len(d) // 2]
    print("%-8s %2d %-44s %7.1f us  (gap before %.1f)" % (tag, i, steps[0][i]["Kernel_Name"].split("(")[0][-44:], d[len(d) // 2], g[len(g) // 2]))
print("%-8s sum of kernel medians %.1f us over %d steps" % (tag, tot, len(steps)))
PY
rm -rf gpurun_out/st_$tag
