"""How busy is the chip in steady state?  From a rocprofv3 --kernel-trace CSV of a bench run: over the middle half of the
trace, the fraction of time with 0 / 1 / 2 / ... kernels running, per-queue busy fractions, and the time per view spent with
only "small" kernels (< SMALL workgroups) running.  `python tools/timeline_stats.py trace.csv`"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    ev.append((s, e, r["Kernel_Name"].split("(")[0][:40], int(r.get("Queue_Id", 0)), wgs))
ev.sort()
t0, t1 = ev[0][0], max(e for _, e, *_ in ev)
lo, hi = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
pts = []
for s, e, name, q, wgs in ev:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        big = wgs >= 1024
        pts.append((s2, 1, big))
        pts.append((e2, -1, big))
pts.sort()
hist = collections.Counter()
nobig = 0
cur = curbig = 0
last = lo
for t, d, big in pts:
    hist[cur] += t - last
    if curbig == 0:
        nobig += t - last
    last = t
    cur += d
    if big:
        curbig += d
hist[cur] += hi - last
if curbig == 0:
    nobig += hi - last
tot = hi - lo
print("window %.1f ms; kernels running at once:" % (tot / 1e6), {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("no kernel of >= 1024 workgroups running: %.3f of the time" % (nobig / tot))
qb = collections.Counter()
for s, e, name, q, wgs in ev:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 > s2:
        qb[q] += e2 - s2
print("busy fraction per queue:", {q: round(v / tot, 3) for q, v in sorted(qb.items())})
nviews = sum(1 for s, e, name, q, wgs in ev if "raster_tile" in name and lo <= s < hi)
print("views in the window:", nviews, "-> %.3f ms per view" % (tot / 1e6 / max(nviews, 1)))
