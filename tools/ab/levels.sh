# per-launch durations of the pyramid kernels of one build: usage levels.sh <name> [ENV=1]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp tools/ab/$1.so orbslam2_amd/liborbfe.so
[ -n "$2" ] && export $2
rm -rf gpurun_out/lv_$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lv_$1 -- python3 bench.py --steps 6 --warmup 2 --cpu-pairs 0 --pipelined 0 --small-batch 0 --natural 0 --host-fed 0 --secondary 0 --no-check > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/lv_$1/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pyr_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive pyramid launches into chains
per = collections.defaultdict(list)
i = 0
chain = []
last = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-28:]
    key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""))
    per[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in per.items():
    v.sort()
    print("$1 $2", k, "n=%d median %.1f us" % (len(v), v[len(v) // 2]))
PY
rm -rf gpurun_out/lv_$1
