#!/bin/bash
# SQ counters of every kvae kernel a python tool launches (counters only, three passes of at
# most eight SQ counters).  usage (on the GPU box): bash tools/pmc_sq.sh NAME tools/some_tool.py [args]
#   -> gpurun_out/pmc_NAME/p*/..., summary on stdout
set -e
NAME=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
mkdir -p $OUT
PROG=$GRAFT_REPO_ROOT/$1; shift
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA -d $OUT/p1 -o p1 --output-format csv -- python3 $PROG "$@" > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/p2 -o p2 --output-format csv -- python3 $PROG "$@" > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/p3 -o p3 --output-format csv -- python3 $PROG "$@" > $OUT/p3.log 2>&1 || echo "pass 3 failed (see $OUT/p3.log)"
python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $(find $OUT -name "*counter_collection.csv")
