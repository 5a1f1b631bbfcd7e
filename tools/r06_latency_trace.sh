#!/bin/bash
# One view at a time under rocprofv3 --kernel-trace: the GPU-side timeline of the LAST view rendered with a second stream
# (static branch on the lane's stream, dynamic branch beside it) and of the last one rendered on one stream -- per kernel its
# start relative to the view's first kernel, its duration, the gap to the previous kernel on the same queue, and per queue the
# sums.  Shows which branch is the critical path of one view alone and what a launch boundary costs there.
# usage (GPU box, through gpurun): bash tools/r06_latency_trace.sh <tag> [bench flags]
tag=${1:-lt}; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/${tag}_trace"
python3 $R/bench.py --latency-only 12 --inflight 1 --warmup 3 "$@" 2>&1 | grep latency-only
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -o k -- python3 $R/bench.py --latency-only 6 --inflight 1 --warmup 3 "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
grep latency-only $R/gpurun_out/${tag}_trace.log
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("pgdvs::", "").replace("void ", "")[:34], r.get("Queue_Id", "?")) for r in rows)
prep = [i for i, e in enumerate(ev) if e[2].startswith("view_prep")]
n = len(prep)
half = 6
def show(lo, hi, label):
    sel = ev[lo:hi]
    t0 = sel[0][0]
    print(f"---- {label}: {len(sel)} kernels, first start -> last end {(max(e[1] for e in sel) - t0) / 1e3:.1f} us")
    lastq = {}
    perq = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for s, e, k, q in sel:
        gap = (s - lastq[q]) / 1e3 if q in lastq else 0.0
        lastq[q] = e
        perq[q][0] += 1; perq[q][1] += (e - s) / 1e3; perq[q][2] += max(gap, 0.0)
        print(f"  q{q:>3s} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {k}")
    for q, (c, d, g) in perq.items():
        print(f"  queue {q}: {c} kernels, kernel time {d:.1f} us, gaps {g:.1f} us")
# the traced run renders 3 warm-up + calibrate ... then 6 views with the second stream, then 6 on one stream
show(prep[-7], prep[-6], "one view, second stream for the dynamic branch (last of its six)")
show(prep[-1], len(ev), "one view, one stream (last of its six)")
PY
find $R/gpurun_out/${tag}_trace -name "*.csv" -size +20M -delete
