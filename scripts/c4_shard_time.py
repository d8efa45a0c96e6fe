import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
N, d, G = 1024, 20, 2000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for rep in range(3):
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=31953150, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G, w["gamma"]); e.synchronize()
    t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
    print(f"C4 shard lanes={e.info()['lanes_per_chain']}: {dt / (G / 10) * 1e6:.2f} us per window", flush=True)
    e.close()
