import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
d, G, K = 5, 2000, 10
for N in (64, 256, 512, 768, 1024, 1536, 2048):
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    out = []
    for lanes in (164, 100):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=5, target=w["target"], lanes_per_chain=lanes)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, w["gamma"]); e.synchronize()
        t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
        out.append((dt / (G / K) * 1e6, e.live_status()[0]))
        e.close()
    print(f"N={N:5d}: wave per chain {out[0][0]:.2f} us per window (LIVE {out[0][1]}), replicated lanes {out[1][0]:.2f} (LIVE {out[1][1]})", flush=True)
