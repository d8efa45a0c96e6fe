#!/bin/bash
# The publisher wave with (rounds 2-3: s_waitcnt vmcnt(0) before every round of stores, build_ab/pubwait.so, -DDEMCZ_PUB_WAIT)
# and without (shipped) the wait for its stores' acknowledgements: C2 / C3 / C4 shard / C5 and N = 2048, twice each, alternating.
# Build the other side first (CPU box):  python scripts/build_variant.py pubwait -DDEMCZ_PUB_WAIT
# -> profiles/r04n_publisher_wait.txt
for i in 1 2; do
for lib in "" "build_ab/pubwait.so"; do
  export DEMCZ_LIB=$lib; [ -z "$lib" ] && unset DEMCZ_LIB
  python bench.py --no-cpu-baseline --no-sweep --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']; print('lib=[$lib]', 'C2 %.3e med %.3e launch %.1f' % (d['value'], d['value_median'], d['roofline']['avg_launch_us']), 'C3 %.2f C4 %.2f C5 %.2f' % (c['C3']['us_per_K_window'], c['C4_shard']['us_per_K_window'], c['C5']['us_per_K_window']))"
  python bench.py --chains-per-gpu 2048 --no-cpu-baseline --no-sweep --no-configs --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   N=2048 %.3e med %.3e launch %.1f' % (d['value'], d['value_median'], d['roofline']['avg_launch_us']))"
done; done
