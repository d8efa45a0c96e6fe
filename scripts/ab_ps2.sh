#!/bin/bash
# A/B of library builds on the C2 launch: slab times at K = 10 (LIVE hand-off, growing archive), the no-hand-off floor at a small
# and at a large archive.  usage: scripts/ab_ps2.sh <lib.so>...   (each line: build, then the three measurements)
for lib in "$@"; do
  echo "== $lib"
  DEMCZ_LIB=$lib python scripts/slab_times.py 2>&1 | grep -v "^$"
  DEMCZ_LIB=$lib python scripts/floor_large_archive.py 2>&1
done
