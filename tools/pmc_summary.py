#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv run: mean counter value per kernel name."""
import csv, glob, sys, collections
d = sys.argv[1]
if len(sys.argv) > 2:  # build id of the liborbfe.so the pass ran on (orbfe_build_id): bench.py replays the file only for that build
    print("# build_id:", sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:40]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}, "n=%d" % len(next(iter(acc[k].values()))))
