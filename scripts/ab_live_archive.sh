#!/bin/bash
# A/B of library builds on the C2 LIVE launch (K = 10) at three archive sizes.  usage: scripts/ab_live_archive.sh <lib.so>...
for lib in "$@"; do
  echo "== $lib"
  for m in 100000 1000000 4000000; do DEMCZ_LIB=$lib python scripts/floor_large_archive.py $m 8 10; done
done
