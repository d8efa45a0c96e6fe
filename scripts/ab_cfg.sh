#!/bin/bash
# scripts/ab_cfg.sh <grep pattern> <gens> lib...: bench_configs rows matching pattern for each library build
pat="$1"; gens="$2"; shift; shift
for lib in "$@"; do echo "== $lib"; DEMCZ_LIB=$PWD/$lib python scripts/bench_configs.py $gens 2>&1 | grep -E "$pat"; done
