#!/bin/bash
# instruction-mix counters of the kernels matching $1 (own pass, no traces beside --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_k
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "$1" --output-format csv -d $R/gpurun_out/pmc_k -o k -- python3 $R/bench.py --steps 4 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $R/gpurun_out/pmc_k.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_k
find $R/gpurun_out/pmc_k -name "*.csv" -delete
