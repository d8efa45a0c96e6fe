"""Floor of a C2 launch's own work at a LARGE archive: K = 1000 (no row hand-off inside a launch) with an initial
archive of M0 rows, so the gathers miss the L2 as they do late in a K = 10 run.  usage: floor_large_archive.py [M0] [slabs] [K]   (K = 10: the LIVE launch itself at that archive size)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, d = 1024, 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
w = demc.workloads.mvnormal_problem(d, N)
rng = np.random.default_rng(0)
Z0 = np.asfortranarray(w["mu"] + 0.1 * rng.standard_normal((M0, d)))
G = S * 1000
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(Z0[-N:], None, Z0)
out = []
for s in range(S):
    e.set_kernel_timing(True)
    e.run(s * 1000 + 1, (s + 1) * 1000, w["gamma"])
    n, ms = e.get_kernel_time()
    out.append(ms * 1e3)
e.close()
print(f"M0={M0} K={K}: us per 1000-generation launch by slab: " + " ".join(f"{v:.0f}" for v in out), flush=True)
