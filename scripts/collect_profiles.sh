#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel trace + the two PMC passes of the
# default bench workload, written under gpurun_out/; scripts/summarize_profiles.py then condenses
# them into profiles/ (tracked).  FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).  The program itself
# follows `--` (python3 <script>): no env / bash -c hop between the profiler and the GPU process.
#   scripts/collect_profiles.sh <tag> [bench flags]            the bench (C2)
#   scripts/collect_profiles.sh <tag> cfg C3|C4|C5 [gens]       one BASELINE config through scripts/run_cfg.py
set -e
tag=${1:-r02}; shift || true
R=$PWD
if [ "$1" = "cfg" ]; then
  prog="$R/scripts/run_cfg.py $2 ${3:-400}"
else
  prog="$R/bench.py --no-cpu-baseline --no-sweep $*"
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $prog > $R/gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $prog > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $prog > $R/gpurun_out/${tag}_write.log 2>&1
echo collected $tag
