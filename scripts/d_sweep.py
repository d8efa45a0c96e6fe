"""bench.py's `configs.d_sweep` on its own (N = 1024, K = 10, d = 5..30 + iso-quad), one line per dimension.
usage: python scripts/d_sweep.py [generations]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import demc_jl_amd as demc

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for r in bench.d_sweep_rows(demc, 31953150, 0, gens=gens):
    if "error" in r:
        print(r)
        continue
    print(f"{r['workload']:62s} {r['kernel']:62s} {r['us_per_K_window_kernels']:6.2f} us/K-window (kernels)  {r['value']:.3e} updates/s  "
          f"{r['roofline']['achieved']:7.1f} {r['roofline']['unit']} = {100 * r['roofline']['frac']:5.2f} % of {r['roofline']['bound']}  live {r['live_launches']} redos {r['live_redos']} launches {r['launches']}  max R-hat {r['max_rhat']:.3f}")
