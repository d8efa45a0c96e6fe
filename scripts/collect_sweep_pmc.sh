#!/bin/bash
# Counter passes of ONE point of bench.py's chain-count sweep (the throughput regime: window_kernel<0, 5, true>, one lane per
# chain) -> gpurun_out/<tag>_N<chains>_{trace,sq1,sq2,fetch,write,tcc}; scripts/pmc_summary.py prints a pass.  Separate passes:
# SQ has 8 slots, TCC 4 (FETCH_SIZE takes 3, WRITE_SIZE 2).  The program itself follows `--`.
#   scripts/collect_sweep_pmc.sh <tag> <chains> [generations]
set -e
tag=${1:-r05}; n=${2:-131072}; g=${3:-200}
R=$PWD
prog="$R/scripts/sweep_point.py $n $g"
o=$R/gpurun_out/${tag}_N${n}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d ${o}_trace -- python3 $prog > ${o}_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d ${o}_sq1 -- python3 $prog > ${o}_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${o}_sq2 -- python3 $prog > ${o}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${o}_fetch -- python3 $prog > ${o}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${o}_write -- python3 $prog > ${o}_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d ${o}_tcc -- python3 $prog > ${o}_tcc.log 2>&1 || echo "tcc pass failed"
echo collected $tag N=$n
