#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/<tag>_stats ; prints the top kernels
tag=${1:-tmp}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.log | cut -c1-300
f=$(find $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-int(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{r["Name"][:60]:60s} {int(r["Calls"]):5d} {int(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:9.2f}')
PY
