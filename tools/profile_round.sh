#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in
# their own --pmc passes (never combined with traces).  Run on the GPU box through gpurun.
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --gnt-rays 0"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o k -- $CMD > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -o k -- $CMD > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -o k -- $CMD > $R/gpurun_out/${tag}_write.log 2>&1
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-200
ls $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write
# drop the bulky per-dispatch traces before the merge back (only the summaries are needed)
find $R/gpurun_out/${tag}_stats -name "*kernel_trace.csv" -delete
