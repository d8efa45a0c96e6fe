"""C4's whole population on one GPU with its full history resident (N = 8192, d = 20, 10 000 generations: 13.1 GB of history,
1.6 GB of archive): allocates, runs, reads the R-hat of the last 1000 generations."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import demc_jl_amd as demc
N, d, G = 8192, 20, 10000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
t0 = time.perf_counter()
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
t1 = time.perf_counter()
e.run(1, G, w["gamma"]); e.synchronize()
t2 = time.perf_counter()
rh = e.rhat(G - 999, G)
print(f"create {t1-t0:.2f} s, run {t2-t1:.3f} s = {N*G/(t2-t1):.3e} updates/s, max R-hat of the last 1000 generations {rh.max():.4f}, archive rows {e.M}, layout {e.info()['lanes_per_chain']}")
e.close()
