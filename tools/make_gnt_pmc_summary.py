#!/usr/bin/env python3
"""profiles/<tag>_gnt_bf16x3_pmc.txt from the passes of tools/pmc_gnt.sh (gpurun_out/pmc_gnt, pmc_gnt2, kt_gnt): the header
table bench.py's gnt_pmc_profile() parses (per kernel: us per dispatch, MFMA / vector wave-instructions, matrix pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x duration)) and every counter's mean per dispatch below it.
usage: make_gnt_pmc_summary.py r06"""
import collections
import csv
import glob
import pathlib
import sys

tag = sys.argv[1]
root = pathlib.Path(__file__).resolve().parent.parent


def short(n):
    return n.split("(")[0].replace("pgdvs::", "").replace("void ", "").strip()


cnt = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("pmc_gnt", "pmc_gnt2"):
    for f in glob.glob(str(root / "gpurun_out" / d / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
dur = collections.defaultdict(list)
for f in glob.glob(str(root / "gpurun_out" / "kt_gnt" / "**" / "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = {k: {c: v / len(ndisp[k][c]) for c, v in cs.items()} for k, cs in cnt.items()}
out = root / "profiles" / f"{tag}_gnt_bf16x3_pmc.txt"
with open(out, "w") as fh:
    fh.write("# rocprofv3 --pmc <8 counters> --kernel-include-regex gnt_ -- python3 tools/gnt_bench.py --rays 1024 --views 24 --stats 1 --iters 2\n")
    fh.write(f"#   (two counter passes + one --kernel-trace pass, tools/pmc_gnt.sh + tools/make_gnt_pmc_summary.py; mean per dispatch; {tag}, committed tree:\n")
    fh.write("#   default product path -- exact bf16x3 products in the view layers and the feed-forward blocks, fp32 matrix instruction elsewhere)\n")
    fh.write("#   kernel                              dispatches   us      (fp32 path)   MFMA insts   all vector insts   matrix pipe busy\n")
    for k in sorted(mean, key=lambda k_: -sum(dur.get(k_, [0]))):
        if not k.startswith("gnt_") or not dur.get(k) or "SQ_VALU_MFMA_BUSY_CYCLES" not in mean[k]:
            continue
        us = sum(dur[k]) / len(dur[k])
        busy = mean[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4e9 * us * 1e-6)
        fh.write(f"#   {k:38s} {us:8.1f}   ( {0.0:7.1f})   {mean[k].get('SQ_INSTS_MFMA', 0) / 1e6:8.1f} M   {mean[k].get('SQ_INSTS_VALU', 0) / 1e6:8.1f} M   {100 * busy:10.1f} %\n")
    fh.write("#   (the fp32-path column is not measured in this pass: 0.0)\n")
    for k in sorted(k_ for k_ in mean if k_.startswith("gnt_")):
        fh.write(f"{k}  dispatches={max(len(s) for s in ndisp[k].values())}\n")
        for c, v in sorted(mean[k].items()):
            fh.write(f"    {c:24s} {v:16.1f}\n")
    fh.write("kernel durations (us): name calls mean total\n")
    for k, v in sorted(((k_, v_) for k_, v_ in dur.items() if k_.startswith("gnt_")), key=lambda kv: -sum(kv[1])):
        fh.write(f"    {k:40s} {len(v):6d} {sum(v) / len(v):10.1f} {sum(v):12.1f}\n")
print("wrote", out)
