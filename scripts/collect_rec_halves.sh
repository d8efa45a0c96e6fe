#!/bin/bash
# does producing a slab's proposal records in pieces small enough to stay in the 256-MiB Infinity Cache between the
# producer's write and the window kernel's read change anything?  The headline (C2) alone, record buffers of 64 (default), 32, 16
# and 8 MiB (DEMCZ_REC_MIB: a launch covers as many generations as one buffer holds), a kernel trace and a FETCH_SIZE pass each (FETCH_SIZE = TCC_EA0_RDREQ x 64 B: the L2's
# memory-side read requests, which per MI355X_MICROARCH.md count Infinity-Cache hits too -- so the time is the evidence, the bytes the check).
# Run on the GPU box from the repo root; scripts/rec_halves_summary.py condenses gpurun_out/<tag>_rec*/ into one table.
set -e
tag=${1:-r05}
R=$PWD
prog="$R/bench.py --no-cpu-baseline --no-sweep --no-configs --steps 20 --warmup 5"
cd /tmp && export TMPDIR=/tmp
for mib in 64 32 16 8; do
  export DEMCZ_REC_MIB=$mib
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_rec${mib}_trace -- python3 $prog > $R/gpurun_out/${tag}_rec${mib}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_rec${mib}_fetch -- python3 $prog > $R/gpurun_out/${tag}_rec${mib}_fetch.log 2>&1
  echo done $mib
done
echo collected
