#!/bin/bash
# Kernel trace + SQ counters of FPMulNode at a small batch (tools/sweep_fused_fpmul.py <elements>), separate passes.
#   bash tools/prof_small_fpmul.sh <outdir-under-gpurun_out> [elements]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=${2:-1024}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/sweep_fused_fpmul.py $N"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/sweep.txt 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -o p -- $B > /dev/null 2>&1 || true
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_means.txt 2>&1 || true
cut -c1-150 $OUT/trace/t_kernel_stats.csv | head -12
cat $OUT/pmc_means.txt
