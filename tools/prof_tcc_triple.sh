#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_tcc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ub -- $GRAFT_REPO_ROOT/tools/ubench_mfma_bfly 20 4 256 16 > /dev/null 2>&1 || echo "ub pass failed"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/bn -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 5 --warmup 2 --no-extra --cpu-sample-log2 10 --prewarm-seconds 0.05 > /dev/null 2>&1 || echo "bench pass failed"
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/r4_tcc/ub | grep -A5 "11, 8, 8" | head -30; python3 tools/pmc_summary.py gpurun_out/r4_tcc/bn | grep -A5 "11, 8, 8" | head -12
