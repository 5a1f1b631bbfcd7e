#!/bin/bash
# steady-state concurrency of the bench loop from a kernel trace (tools/timeline_stats.py); $@ = extra bench flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o k -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep "$@" > $R/gpurun_out/tl.log 2>&1
tail -1 $R/gpurun_out/tl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames/s under the profiler', d['value'], d['config'].get('views_in_flight'))"
python3 $R/tools/timeline_stats.py $(find $R/gpurun_out/tl -name "*kernel_trace.csv")
find $R/gpurun_out/tl -name "*kernel_trace.csv" -delete
