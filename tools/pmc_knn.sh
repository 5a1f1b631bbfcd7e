#!/bin/bash
# instruction-mix counters of the kNN query kernel (own pass, no traces beside --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_knn
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-include-regex "grid_query" --output-format csv -d $R/gpurun_out/pmc_knn -o k -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $R/gpurun_out/pmc_knn.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_knn
