"""Fixed vs per-generation cost of the window kernel: time runs with different K (generations per
launch) on the C2 workload.  usage: python scripts/probe_window_cost.py [lanes ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

N, d, G = 1024, 5, 4000
w = demc.workloads.mvnormal_problem(d, N)
for lanes in [int(a) for a in sys.argv[1:]] or [1, 8]:
    for K in (1, 2, 5, 10, 20, 50, 200, 1000):
        M0 = w["Zinit"].shape[0]
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=[range(d)],
                           eps_scale=w["eps_scale"], seed=1, target=w["target"], lanes_per_chain=lanes)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, 2.38); e.synchronize()
        t0 = time.perf_counter()
        e.run(G + 1, 2 * G, 2.38); e.synchronize()
        dt = time.perf_counter() - t0
        nl = G // K
        print(f"lanes={lanes} K={K:5d} launches={nl:5d} total={dt*1e3:8.2f} ms  per-launch={dt/nl*1e6:8.2f} us  per-gen={dt/G*1e6:6.3f} us", flush=True)
        e.close()
