#!/bin/bash
# (the command pins the lane count: under the profiler the host is slow and a lane probe would measure the profiler)
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc
# passes, then two instruction-mix passes restricted to the co-dominant kernels (never combined with traces).
# Run on the GPU box through gpurun; tools/make_profile_summary.py condenses the output into profiles/.
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 10 --warmup 2 --inflight 3 --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep"
PMCCMD="python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --no-kernel-timing --gnt-rays 0 --no-scene-sweep"
KERNELS="raster_tile|grid_query_tpq|agg_push|agg_step|agg_rows|agg_select|raster_fill|dyn_splat_scatter"
cd /tmp && export TMPDIR=/tmp
for d in stats fetch write pmc1 pmc2; do rm -rf $R/gpurun_out/${tag}_$d; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o k -- $CMD > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -o k -- $CMD > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_write -o k -- $CMD > $R/gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --kernel-include-regex "$KERNELS" --output-format csv -d $R/gpurun_out/${tag}_pmc1 -o k -- $PMCCMD > $R/gpurun_out/${tag}_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD \
  --kernel-include-regex "$KERNELS" --output-format csv -d $R/gpurun_out/${tag}_pmc2 -o k -- $PMCCMD > $R/gpurun_out/${tag}_pmc2.log 2>&1
tail -1 $R/gpurun_out/${tag}_stats.log | cut -c1-200
ls $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_fetch $R/gpurun_out/${tag}_write $R/gpurun_out/${tag}_pmc1 $R/gpurun_out/${tag}_pmc2
# which runtime copies / fills the loop issues (grid size, stream, calls): they are not launches of this library
python3 - $R/gpurun_out/${tag}_stats/k_kernel_trace.csv > $R/gpurun_out/${tag}_stats/runtime_copies.txt 2>&1 <<'PY'
import collections, csv, sys
c = collections.Counter()
n = 0
for r in csv.DictReader(open(sys.argv[1])):
    n += 1
    if "rocclr" in r["Kernel_Name"]:
        c[(r["Kernel_Name"][:40], r["Grid_Size_X"], r["Stream_Id"])] += 1
print(n, "dispatches")
for k, v in c.most_common(30):
    print(v, k)
PY
# drop the bulky per-dispatch traces before the merge back (only the summaries are needed)
find $R/gpurun_out/${tag}_stats -name "*kernel_trace.csv" -delete
