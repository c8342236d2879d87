#!/bin/bash
# Kernel trace + PMC passes for the bench workload (separate passes; no tracing domains mixed in).  Run on the GPU box:
#   bash tools/prof_pmc.sh <outdir-under-gpurun_out> [extra bench args]
# The trace pass runs bench.py exactly as the driver does (--steps 20 --warmup 5); the counter passes shorten the
# pre-warm and the CPU legs (counters are per launch).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extra --cpu-sample-log2 12 $@"
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extra --cpu-sample-log2 10 --prewarm-seconds 0.05 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $T > $OUT/trace.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- $B > /dev/null 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $B > /dev/null 2>&1 || true
find $OUT -name "*.csv" | head -30
