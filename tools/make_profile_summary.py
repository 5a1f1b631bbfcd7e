#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/rNN_{stats,fetch,write}) into the tracked
profiles/ summaries:
  profiles/rNN_kernel_stats.csv   per-kernel calls / total / mean duration (kernel-trace --stats)
  profiles/rNN_hbm_traffic.json   per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
                                   (separate --pmc passes; FETCH_SIZE doubled as
                                   MI355X_MICROARCH.md 'HBM' prescribes for gfx950 -- checked
                                   here on dyn_splat_finish, whose 66.4 MB of coalesced dword
                                   reads are reported as 33.2 MB)
usage: make_profile_summary.py r01 "<bench command that was profiled>"
"""
import collections
import csv
import glob
import json
import pathlib
import sys

tag = sys.argv[1]
cmd = sys.argv[2] if len(sys.argv) > 2 else ""
root = pathlib.Path(__file__).resolve().parent.parent
out = root / "profiles"
out.mkdir(exist_ok=True)


def short(n):
    return n.split("(")[0].replace("pgdvs::", "").replace("void ", "").strip()[:90]


def counter(dirname, cname):
    agg, disp = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(str(root / "gpurun_out" / dirname / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            agg[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: (v / len(disp[k]), len(disp[k])) for k, v in agg.items()}


rows = []
for f in glob.glob(str(root / "gpurun_out" / f"{tag}_stats" / "**" / "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((short(r["Name"]), int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
rows.sort(key=lambda r: -r[2])
with open(out / f"{tag}_kernel_stats.csv", "w") as fh:
    fh.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
    fh.write("kernel,calls,total_us,avg_us,percent\n")
    for n, c, t, a, p in rows:
        fh.write(f"{n},{c},{t/1e3:.1f},{a/1e3:.2f},{p:.2f}\n")

fetch, write = counter(f"{tag}_fetch", "FETCH_SIZE"), counter(f"{tag}_write", "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fk, n = fetch.get(k, (0.0, 0))
    wk, n2 = write.get(k, (0.0, 0))
    traffic[k] = {"launches_profiled": max(n, n2), "FETCH_SIZE_KB_raw": round(fk, 1), "WRITE_SIZE_KB_raw": round(wk, 1),
                  "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
# per view: launches per view from the kernel-trace pass (one raster_tile launch = one view)
calls = {n: c for n, c, _, _, _ in rows}
views = max([c for n, c in calls.items() if n.startswith("raster_tile_kernel")] or [0])
total = 0
if views:
    for k, v in traffic.items():
        v["launches_per_view"] = round(calls.get(k, v["launches_profiled"]) / views, 2)
        v["hbm_bytes_per_view"] = int(v["hbm_bytes_per_launch"] * calls.get(k, v["launches_profiled"]) / views)
        total += v["hbm_bytes_per_view"]
json.dump({"command": cmd, "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 FETCH_SIZE correction); per view = x launches "
                                   "per view of the kernel-trace pass of the same command",
           "views_profiled": views, "total_hbm_bytes_per_view": total,
           "kernels": traffic}, open(out / f"{tag}_hbm_traffic.json", "w"), indent=1)
print("wrote", out / f"{tag}_kernel_stats.csv", out / f"{tag}_hbm_traffic.json")
