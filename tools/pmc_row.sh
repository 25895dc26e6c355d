#!/bin/bash
# SQ counters of the 16-lanes-per-sequence objective kernel (counters only, two passes).
# usage (on the GPU box): bash tools/pmc_row.sh [B]   -> gpurun_out/pmc_row/*.csv + summary
set -e
B=${1:-4096}
export TMPDIR=/tmp KVAE_SEQ_LANES=16
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_row
mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_VALU_TRANS -d $OUT/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_loss_kernels.py $B > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_loss_kernels.py $B > $OUT/p2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $(find $OUT -name "*counter_collection.csv") | tee $OUT/summary.txt
