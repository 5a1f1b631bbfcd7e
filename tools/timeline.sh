#!/bin/bash
# rocprofv3 kernel trace of a short bench run, reduced on the box to a timeline summary: for the timed
# region, GPU busy time (union of kernel intervals), per-kernel wall share when running concurrently
# and idle gaps.  Run on the GPU box through gpurun: tools/timeline.sh <tag> [bench flags]
tag=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/${tag}_trace"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -o k -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing --gnt-rays 0 "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
tail -1 $R/gpurun_out/${tag}_trace.log | cut -c1-160
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("pgdvs::", "")[:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
# the timed region: the last views of the run = the last agg_finalize kernels (one per view) and everything between
# them -- 24 of them when the trace holds that many (the command above times 30 steps), otherwise all but the first
fin = [i for i, e in enumerate(ev) if e[2].startswith("agg_finalize")]
if len(fin) < 3:
    sys.exit(f"only {len(fin)} agg_finalize kernels in the trace: nothing to reduce (pass more --steps)")
n_views = min(24, len(fin) - 1)
lo = ev[fin[-n_views - 1]][1]
hi = ev[fin[-1]][1]
sel = [e for e in ev if e[0] >= lo and e[1] <= hi]
span = (hi - lo) / 1e3
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{n_views} views in {span:.0f} us = {span / n_views:.1f} us/view; some kernel running {busy / 1e3 / n_views:.1f} us/view; idle {(span - busy / 1e3) / n_views:.1f} us/view")
# per kernel: summed duration and the time during which it was the ONLY kernel running
tot = collections.Counter(); cnt = collections.Counter()
for s, e, k, _ in sel: tot[k] += e - s; cnt[k] += 1
# sweep line: attribute each time slice equally to the kernels running in it
pts = sorted([(s, 1, k) for s, e, k, _ in sel] + [(e, -1, k) for s, e, k, _ in sel])
share = collections.Counter(); active = collections.Counter(); last = None; conc = collections.Counter()
for t, d, k in pts:
    if last is not None and t > last:
        n = sum(active.values())
        if n:
            for kk, c in active.items():
                if c: share[kk] += (t - last) * c / n
        conc[min(n, 6)] += t - last
    active[k] += d; last = t
print("concurrency histogram (us/view):", {k: round(v / 1e3 / n_views, 1) for k, v in sorted(conc.items())})
print(f"{'kernel':40s} {'n/view':>6s} {'sum us':>8s} {'avg us':>8s} {'share us':>9s}")
for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:24]:
    print(f"{k:40s} {cnt[k] / n_views:6.1f} {tot[k] / 1e3 / n_views:8.1f} {tot[k] / 1e3 / cnt[k]:8.2f} {v / 1e3 / n_views:9.1f}")
PY
find $R/gpurun_out/${tag}_trace -name "*.csv" -size +20M -delete
