#!/bin/bash
# Two chains to a wave (N = 2048, 1536): depth of the history ring between the chain waves and the publisher wave, 4 passes (shipped
# until round 5) against 8 and 16.  The stamps (profiles/r05_ps2d_stamps_2048.txt) show 1.6 polls of a FULL ring per pass: a pass
# leaves the ring when all four chain waves have posted it, so the ring's depth is how far a workgroup's waves may drift apart.
# Build first (CPU box):  python scripts/build_variant.py hs8 -DPS2_HSLOTS_N=8 ; python scripts/build_variant.py hs16 -DPS2_HSLOTS_N=16
for i in 1 2 3; do
for lib in "" "build_ab/hs8.so" "build_ab/hs16.so"; do
  export DEMCZ_LIB=$lib; [ -z "$lib" ] && unset DEMCZ_LIB
  for n in 2048 1536; do
  python bench.py --chains-per-gpu $n --no-cpu-baseline --no-sweep --no-configs --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=[$lib] N=$n %.3e med %.3e launch %.1f us' % (d['value'], d['value_median'], d['roofline']['avg_launch_us']))"
  done
done; done
