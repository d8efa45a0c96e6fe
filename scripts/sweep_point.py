"""One point of bench.py's chain-count sweep: python scripts/sweep_point.py N [gens] [d]   (DEMCZ_LIB: another build of the library)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
import demc_jl_amd as demc
n = int(sys.argv[1]); g = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for _ in range(2):
    p = bench.throughput_point(demc, n, d, 10, 31953150, g, 0)
    print({k: (f"{v:.4g}" if isinstance(v, float) else v) for k, v in p.items()}, flush=True)
