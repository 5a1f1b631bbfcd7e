#!/bin/bash
# HBM traffic of selected kernels (separate FETCH_SIZE / WRITE_SIZE passes, never combined with traces): bash tools/r05_traffic_kernels.sh TAG "regex"
tag=${1:-t}; re=${2:-agg_step|agg_rows}
R=$GRAFT_REPO_ROOT
PMCCMD="python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${tag}_$c
  rocprofv3 --pmc $c --kernel-include-regex "$re" --output-format csv -d $R/gpurun_out/${tag}_$c -o k -- $PMCCMD > $R/gpurun_out/${tag}_$c.log 2>&1
done
python3 - $R/gpurun_out/${tag} <<'PY'
import collections, csv, glob, sys
base = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, set()])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob(base + "_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].replace("pgdvs::", "").replace("void ", "")[:60]
                tot[k][i] += float(r["Counter_Value"])
                tot[k][2].add(r["Dispatch_Id"])
for k, (fe, wr, d) in tot.items():
    n = max(len(d), 1)
    print(f"{k:50s} launches {n:5d}  fetch {2 * fe / n / 1024:9.2f} MB (x2 gfx950)  write {wr / n / 1024:9.2f} MB  total {(2 * fe + wr) / n / 1024:9.2f} MB per launch")
PY
