import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
N, d, S = 1024, 5, 8
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for K in (5, 10, 20, 50, 100, 250, 1000):
    G = S * 1000
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    out = []
    for s in range(S):
        e.set_kernel_timing(True)
        e.run(s * 1000 + 1, (s + 1) * 1000, w["gamma"])
        n, ms = e.get_kernel_time()
        out.append(ms * 1e3)
    e.close()
    print(f"K={K:5d}: " + " ".join(f"{v:.0f}" for v in out), flush=True)
