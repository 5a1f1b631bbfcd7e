#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc / --kernel-trace CSV output per kernel (mean per dispatch)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("pgdvs::", "").replace("void ", "")
        if filt and filt not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        n = len(ndisp[k])
        print(f"{k}  dispatches={n}")
        for c, val in sorted(v.items()):
            print(f"    {c:24s} {val / n:16.1f}")
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("pgdvs::", "").replace("void ", "")
        if filt and filt not in k:
            continue
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("kernel durations (us): name calls mean total")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"    {k:40s} {len(v):6d} {sum(v)/len(v):10.1f} {sum(v):12.1f}")
