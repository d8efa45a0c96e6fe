import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import demc_jl_amd as demc
N, d, G = int(sys.argv[1]), 20, 300
w = demc.workloads.mvnormal_problem(d, N)
blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=0, blockindex=blocks, eps_scale=w["eps_scale"], seed=5, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G, w["gamma"]); e.synchronize()
t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
print(N, e.info()["lanes_per_chain"], f"{N*G/dt:.3e} updates/s")
