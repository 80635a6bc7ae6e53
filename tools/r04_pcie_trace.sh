#!/bin/bash
# timeline of the host-fed lanes: how busy is the upload engine, what runs beside what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for mode in direct unpacked; do
rm -rf gpurun_out/pt_$mode
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/pt_$mode -- python3 tools/pcie_trace.py $mode 4 24 > /dev/null 2> gpurun_out/r04/pt_$mode.err || { tail -5 gpurun_out/r04/pt_$mode.err; exit 1; }
python3 - $mode <<'PY'
import csv, glob, sys
mode = sys.argv[1]
mc = sorted(csv.DictReader(open(glob.glob("gpurun_out/pt_%s/*/*memory_copy_trace.csv" % mode)[0])), key=lambda r: int(r["Start_Timestamp"]))
kt = sorted(csv.DictReader(open(glob.glob("gpurun_out/pt_%s/*/*kernel_trace.csv" % mode)[0])), key=lambda r: int(r["Start_Timestamp"]))
print(mode, "columns", list(mc[0].keys()))
big = [r for r in mc if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 200000]
ups = [r for r in mc if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300000]
ups = ups[-24:]
t0, t1 = int(ups[0]["Start_Timestamp"]), int(ups[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ups)
print(mode, "uploads %d  span %.2f ms  busy %.2f ms  mean %.3f ms  -> %.0f pairs/s over the span" % (len(ups), (t1 - t0) / 1e6, busy / 1e6, busy / len(ups) / 1e6, 64 * len(ups) / ((t1 - t0) / 1e9)))
durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in ups)
print(mode, "upload ms min/med/max", durs[0], durs[len(durs) // 2], durs[-1])
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6 for a, b in zip(ups, ups[1:])]
print(mode, "gaps between uploads ms:", [round(g, 3) for g in gaps])
dn = [r for r in mc if "DEVICE_TO_HOST" in r["Direction"] and int(r["Start_Timestamp"]) >= t0]
print(mode, "downloads in span:", len(dn), "total ms %.2f" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in dn) / 1e6))
pk = [r for r in kt if "pack_results" in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= t0]
if pk:
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in pk)
    print(mode, "pack kernel ms min/med/max", d[0], d[len(d) // 2], d[-1])
kb = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1)
print(mode, "kernel time inside the span %.2f ms" % (kb / 1e6))
PY
rm -rf gpurun_out/pt_$mode
done
