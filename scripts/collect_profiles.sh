#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: kernel trace + the two PMC passes of the
# default bench workload, written under gpurun_out/; scripts/summarize_profiles.py then condenses
# them into profiles/ (tracked).  FETCH_SIZE and WRITE_SIZE need separate passes (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
tag=${1:-r01}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_write -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_write.log 2>&1
echo collected $tag
