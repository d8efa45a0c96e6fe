#!/bin/bash
# SQ counter passes of the bench (or of one config through run_cfg.py) -> gpurun_out/<tag>_sq{1,2}; scripts/pmc_summary.py prints them.
#   scripts/collect_sq.sh <tag> [bench flags]   |   scripts/collect_sq.sh <tag> cfg C3|C4|C5 [gens]
set -e
tag=${1:-r03}; shift || true
R=$PWD
if [ "$1" = "cfg" ]; then prog="$R/scripts/run_cfg.py $2 ${3:-400}"; else prog="$R/bench.py --no-cpu-baseline --no-sweep $*"; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_sq1 -- python3 $prog > $R/gpurun_out/${tag}_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_sq2 -- python3 $prog > $R/gpurun_out/${tag}_sq2.log 2>&1
echo collected sq $tag
