import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
for d in (8, 10):
    N, G = 1024, 2000
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    res = {}
    for lanes in (164, 100):
        e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=5, target=w["target"], lanes_per_chain=lanes)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, w["gamma"]); e.synchronize()
        t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
        ch, lo = e.get_history(1, 2 * G)
        res[lanes] = (ch, lo)
        print(f"d={d} lanes={e.info()['lanes_per_chain']}: {dt / (G / 10) * 1e6:.2f} us per window  live={e.live_status()}", flush=True)
        e.close()
    print("  identical:", np.array_equal(res[164][0], res[100][0]) and np.array_equal(res[164][1], res[100][1]))
