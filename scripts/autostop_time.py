"""Cost of the autostop decision: demcz_run_checked with a threshold that never triggers vs monitoring mode (C2, 10 slabs)."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
N, d, G = 1024, 5, 11000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for thr in (0.0, 1e-9, 0.0, 1e-9):
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run_checked(1, 1000, 2.38, 1000, 0.0); e.synchronize()
    t0 = time.perf_counter()
    gs, mx, _ = e.run_checked(1001, G, 2.38, 1000, thr); e.synchronize()
    dt = time.perf_counter() - t0
    print(f"threshold {thr:g}: {dt*1e3:.3f} ms for 10000 generations -> {N*10000/dt:.3e} upd/s ({dt/10*1e6:.0f} us per slab), checks {len(mx)}")
    e.close()
