#!/bin/bash
# A/B the C2 probe over several builds of the HIP library: scripts/ab.sh "<lanes>" lib1.so lib2.so ...
lanes="$1"; shift
for lib in "$@"; do
  echo "== $lib"
  DEMCZ_LIB=$PWD/$lib python scripts/probe_window_cost.py $lanes 2>&1 | grep -E "K=   10|K= 1000"
done
