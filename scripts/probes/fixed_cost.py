"""Per-slab and per-call cost of demcz_run_checked at C2 (monitoring, 1000-generation slabs): wall time of calls of 10, 20, 40 slabs."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import demc_jl_amd as demc
N, d, K = 1024, 5, 10
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
G = 1000 * 120
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
g = 1
e.run_checked(g, g + 4999, w["gamma"], 1000, 0.0); e.synchronize(); g += 5000
res = {}
for slabs in (10, 20, 40, 20, 10):
    t0 = time.perf_counter()
    e.run_checked(g, g + 1000 * slabs - 1, w["gamma"], 1000, 0.0)
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    g += 1000 * slabs
    res.setdefault(slabs, []).append(t2 - t0)
    print(f"{slabs:3d} slabs: call returns after {1e3*(t1-t0):.3f} ms, synchronised after {1e3*(t2-t0):.3f} ms = {1e6*(t2-t0)/slabs:.1f} us per slab")
per = (min(res[40]) - min(res[10])) / 30
print(f"per slab {1e6*per:.1f} us; per call {1e6*(min(res[20]) - 20 * per):.0f} us")
e.close()
