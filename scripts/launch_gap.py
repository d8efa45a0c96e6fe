import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
S = 20
N, d, K, every = 1024, 5, 10, 1000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
G = (S + 5) * every
for mode in ("run per slab, no checks", "run_checked monitor", "one run call of 20000 gens"):
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, 5 * every, w["gamma"]); e.synchronize()
    t0 = time.perf_counter()
    if mode.startswith("run per"):
        for s in range(S):
            e.run((5 + s) * every + 1, (6 + s) * every, w["gamma"])
    elif mode.startswith("run_checked"):
        e.run_checked(5 * every + 1, G, w["gamma"], every, 0.0)
    else:
        e.run(5 * every + 1, G, w["gamma"])
    e.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:32s}: {dt / S * 1e6:7.1f} us per 1000 generations (wall), launches {e.info()['window_launches']}", flush=True)
    e.close()
