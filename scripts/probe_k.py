"""Per-launch fixed cost vs per-generation cost for a config: python scripts/probe_k.py C5|C3|C2|C4 [lanes]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
name = sys.argv[1]; lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = 400
if name == "C5": N, d = 2048, 10; w = demc.workloads.linreg_problem(d, N); blocks = [range(d)]
elif name == "C3": N, d = 4096, 20; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
elif name == "C4": N, d = 1024, 20; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(d)]
else: N, d = 1024, 5; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(d)]
for K in (1, 2, 10, 50, 200):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=blocks, eps_scale=w["eps_scale"], seed=1,
                       target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G, w["gamma"]); e.synchronize()
    t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
    print(f"{name} lanes={e.info()['lanes_per_chain']} K={K:4d} per-launch={dt/(G/K)*1e6:9.2f} us per-gen={dt/G*1e6:8.3f} us", flush=True)
    e.close()
