"""What a LIVE launch of window_kernel_ps2 costs BEFORE any boundary is counted (round 5): C2's shape, 8 slabs of 1000 generations at a
given K -- K = 1000: no boundary inside a launch, the non-LIVE instantiation; K = 250 / 500: the LIVE instantiation with 4 / 2
boundaries a launch.  Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations (DEMCZ_PRODUCE_SERIAL=1: the producer
in front of the consumer instead of beside it).   usage: python scripts/live_fixed_cost.py <K> [M0]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
K = int(sys.argv[1])
N, d, S = 1024, 5, 8
w = demc.workloads.mvnormal_problem(d, N)
Z0 = w["Zinit"]
if len(sys.argv) > 2:
    M0 = int(sys.argv[2])
    rng = np.random.default_rng(1)
    Z0 = np.asfortranarray(np.vstack([rng.standard_normal((M0 - Z0.shape[0], d)) * 0.1 + w["mu"], Z0]))
M0 = Z0.shape[0]
G = S * 1000
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(Z0[-N:], None, Z0)
out = []
for s in range(S):
    e.set_kernel_timing(True)
    e.run(s * 1000 + 1, (s + 1) * 1000, w["gamma"])
    n, ms = e.get_kernel_time()
    out.append(ms * 1e3)
print(f"K={K} M0={M0} {e.kernel_name()}: " + " ".join(f"{v:.0f}" for v in out), flush=True)
e.close()
