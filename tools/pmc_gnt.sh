#!/bin/bash
# MFMA / issue / wait counters of the GNT kernels (own pass, no traces beside --pmc) + kernel durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gnt $R/gpurun_out/pmc_gnt2 $R/gpurun_out/kt_gnt
ARGS="--rays ${RAYS:-1024} --views ${VIEWS:-24} --stats 1 --iters 2"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "gnt_" --output-format csv -d $R/gpurun_out/pmc_gnt -o k -- python3 $R/tools/gnt_bench.py $ARGS > $R/gpurun_out/pmc_gnt.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-include-regex "gnt_" --output-format csv -d $R/gpurun_out/pmc_gnt2 -o k -- python3 $R/tools/gnt_bench.py $ARGS >> $R/gpurun_out/pmc_gnt.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "gnt_" --output-format csv -d $R/gpurun_out/kt_gnt -o k -- python3 $R/tools/gnt_bench.py $ARGS >> $R/gpurun_out/pmc_gnt.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_gnt gnt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_gnt2 gnt
python3 $R/tools/pmc_summary.py $R/gpurun_out/kt_gnt gnt
