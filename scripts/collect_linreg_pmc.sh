#!/bin/bash
# Counter passes of the regression target at the reference example's shape (scripts/linreg_d26.py 100 1000 0: window_kernel_ml<LINREG_SSE, 26, 16, ..., COOP>)
# -> gpurun_out/<tag>_lr_{sq1,sq2,fetch,tcc}; scripts/pmc_summary.py <dir> window_kernel_ml prints a pass.  Separate passes; the program follows `--`.
set -e
tag=${1:-r05}
R=$PWD
prog="$R/scripts/linreg_d26.py 100 1000 0"
o=$R/gpurun_out/${tag}_lr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d ${o}_sq1 -- python3 $prog > ${o}_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${o}_sq2 -- python3 $prog > ${o}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${o}_fetch -- python3 $prog > ${o}_fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d ${o}_tcc -- python3 $prog > ${o}_tcc.log 2>&1 || echo "tcc pass failed"
echo collected $tag linreg
