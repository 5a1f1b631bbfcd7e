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
            disp[k].add((f, r["Dispatch_Id"]))
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

# ---- instruction-mix counters of the co-dominant kernels (gpurun_out/<tag>_pmc1, _pmc2: separate --pmc passes of
# `bench.py --steps 4 --warmup 1 --inflight 1 --no-side-stream ...` restricted to those kernels) ->
#   profiles/<tag>_pmc_instructions.json  (read by bench.py for the VALU-issue figure)
#   profiles/<tag>_pmc_<kernel>.txt       (one text block per kernel, every counter, mean per launch)
pmc = collections.defaultdict(dict)
for d in (f"{tag}_pmc1", f"{tag}_pmc2"):
    # one pass = one directory: sums and dispatch sets are merged over ALL of its files (one per process / agent) before
    # the division, so that the means do not depend on the order in which glob returns them
    agg, disp = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for f in sorted(glob.glob(str(root / "gpurun_out" / d / "**" / "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add((f, r["Dispatch_Id"]))
    for k, v in agg.items():
        for c, val in v.items():
            pmc[k][c] = val / len(disp[k])
        pmc[k]["launches_profiled"] = max(pmc[k].get("launches_profiled", 0), len(disp[k]))
if pmc:
    import re
    # several instantiations of one kernel (raster_tile<3, 2048, ..> per view, <3, 4096, ..> for an occasional long list):
    # the one launched most often stands for the name
    keyed = {}
    for k, cs in sorted(pmc.items(), key=lambda kv: kv[1].get("launches_profiled", 0)):
        keyed[re.sub(r"<.*>", "", k).replace("_kernel", "")] = {c: round(v, 1) for c, v in sorted(cs.items())}
    # (agg_push_kernel<1024> is frame 0's launch: bench.py names it agg_push0)
    if "agg_push" in keyed:
        keyed["agg_push0"] = keyed.pop("agg_push")
    json.dump({"command": "rocprofv3 --pmc <8 SQ counters> --kernel-include-regex <co-dominant kernels> -- python3 bench.py --steps 4 "
                          "--warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep (two passes)",
               "note": "mean per launch; SQ_INSTS_* are wave-instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles "
                       "summed over waves (MI355X_MICROARCH.md); issue cost per wave-instruction and SIMD: profiles/r03_valu_rate.txt",
               "kernels": keyed}, open(out / f"{tag}_pmc_instructions.json", "w"), indent=1)
    names = {"raster_tile": "raster_tile", "grid_query_tpq": "knn_tpq", "agg_push0": "agg_push0", "agg_step": "agg_step",
             "agg_rows": "agg_rows", "dyn_splat_scatter": "dyn_splat_scatter"}
    for k, cs in keyed.items():
        with open(out / f"{tag}_pmc_{names.get(k, k)}.txt", "w") as fh:
            fh.write(f"# {k}: rocprofv3 --pmc counters, mean per launch ({int(cs.get('launches_profiled', 0))} launches x XCDs profiled)\n")
            for c, v in cs.items():
                if c != "launches_profiled":
                    fh.write(f"{c:28s} {v:16.1f}\n")
            if cs.get("SQ_INSTS_VALU") and cs.get("SQ_WAVES"):
                fh.write(f"{'VALU insts per wave':28s} {cs['SQ_INSTS_VALU'] / cs['SQ_WAVES']:16.1f}\n")
                fh.write(f"{'SALU insts per wave':28s} {cs.get('SQ_INSTS_SALU', 0) / cs['SQ_WAVES']:16.1f}\n")
    print("wrote", out / f"{tag}_pmc_instructions.json")
