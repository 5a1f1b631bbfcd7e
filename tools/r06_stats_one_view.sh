#!/bin/bash
# rocprofv3 --kernel-trace --stats of the benchmark with ONE view in flight on ONE stream (the mode in which bench.py takes its
# per-kernel HIP-event times: no kernel of the view runs beside another) -> gpurun_out/<tag>_stats; condensed by
# tools/make_profile_summary.py <tag> into profiles/<tag>_kernel_stats.csv.  The three-lane passes of tools/profile_round.sh
# show what the same kernels take beside each other.
tag=${1:-r06iso}
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 10 --warmup 2 --inflight 1 --no-side-stream --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o k -- $CMD > $R/gpurun_out/${tag}_stats.log 2>&1
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-200
find $R/gpurun_out/${tag}_stats -name "*kernel_trace.csv" -delete
