#!/bin/bash
# scripts/ab_bench.sh "<bench args>" lib1.so lib2.so ... : one bench line per build of the HIP library
args="$1"; shift
for lib in "$@"; do
  DEMCZ_LIB=$PWD/$lib python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['config']['chains_total'], 'L=%d'%d['config']['lanes_per_chain'], '%.3e upd/s'%d['value'], 'launch_us=%.1f'%r['avg_launch_us'], 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']))"
done
