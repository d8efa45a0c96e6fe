#!/bin/bash
# Run on the GPU box from the repo root: FETCH_SIZE of the calibration probe (scripts/probes/fetch_calibration.hip: the consumer's
# two read patterns and a plain stream, each with a known byte count) -> gpurun_out/<tag>_cal/, and the probe's own line of known
# bytes -> gpurun_out/<tag>_cal.json.  scripts/summarize_profiles.py <tag> reads both.
set -e
tag=${1:-r04}
R=$PWD
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/fetch_cal $R/scripts/probes/fetch_calibration.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_cal -- /tmp/fetch_cal > $R/gpurun_out/${tag}_cal.json 2> $R/gpurun_out/${tag}_cal.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_cal_trace -- /tmp/fetch_cal > /dev/null 2>&1
echo calibration collected $tag
