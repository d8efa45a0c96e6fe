"""us per generation at small K (a pass / chunk never spans a K boundary): one wave per chain against eight replicated lanes.
usage: python scripts/k_small.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
N, d, G = 1024, 5, 2000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for K in (1, 2, 3, 4, 5, 7, 10):
    out = []
    for lanes in (164, 100):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=5, target=w["target"], lanes_per_chain=lanes)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, w["gamma"]); e.synchronize()
        t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
        out.append(dt / G * 1e6)
        e.close()
    print(f"K={K:3d}: wave per chain {out[0]:.3f} us per generation, replicated lanes {out[1]:.3f}", flush=True)
