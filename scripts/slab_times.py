"""Window-kernel time per 1000-generation slab of the C2 workload as the archive grows (hand-off waits shrink with
N/M), next to the same launches with K = 1000 (no hand-off at all) -- the floor of the launch's own work.
usage: python scripts/slab_times.py [slabs] [N] [d]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 25
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
import os
NOHIST = bool(os.environ.get("NOHIST"))       # NOHIST=1: no chain / log_obj history kept (what the history stores cost)
for K in (10, 1000):
    G = S * 1000
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=0 if NOHIST else G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    out = []
    for s in range(S):
        e.set_kernel_timing(True)
        e.run(s * 1000 + 1, (s + 1) * 1000, w["gamma"])
        n, ms = e.get_kernel_time()
        out.append(ms * 1e3)
    e.close()
    print(f"K={K:5d} N={N} d={d}: us per 1000-generation launch by slab: " + " ".join(f"{v:.0f}" for v in out), flush=True)
    print(f"        mean of slabs 6..{S}: {np.mean(out[5:]):.1f} us = {np.mean(out[5:]) / 100:.3f} us per 10 generations", flush=True)
