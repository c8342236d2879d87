#!/bin/bash
# usage: bash tools/prof_any.sh <outdir> <python script + args>   (PMC passes + kernel trace)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.out 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $B > /dev/null 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B > /dev/null 2>&1 || true
