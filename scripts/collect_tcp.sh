#!/bin/bash
# TCP (vector L1 + its TLB) counter passes of one C2 LIVE run at a given archive size -> gpurun_out/<tag>_tcp{1..4};
# scripts/pmc_summary.py prints them (four TCP counters fit one pass).   usage: scripts/collect_tcp.sh <tag> <M0> [slabs]
tag=$1; m0=$2; slabs=${3:-6}
R=$PWD
prog="$R/scripts/floor_large_archive.py $m0 $slabs 10"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_tcp$i -- python3 $prog > $R/gpurun_out/${tag}_tcp$i.log 2>&1 || echo "pass $i failed"
done
echo collected tcp $tag
