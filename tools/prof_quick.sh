#!/bin/bash
# durations of consecutive dispatches of one kernel in a short single-lane bench run; $1 = name filter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/q_stats
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/q_stats -o k -- python3 $R/bench.py --steps 4 --warmup 2 --inflight 1 --launch eager --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $R/gpurun_out/q_stats.log 2>&1
python3 - "$1" <<'PY'
import csv,glob,os,sys
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/q_stats/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(len(d), "dispatches; last 24 (us):", " ".join(f"{x:.0f}" for x in d[-24:]))
PY
find $R/gpurun_out/q_stats -name "*kernel_trace.csv" -delete
