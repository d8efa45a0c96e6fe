"""Layout crossover in the chain count: us per K-window for each layout at dimension d.
usage: python scripts/crossover.py d N1 N2 ...   (layouts tried: split (100), 8/16 lanes, 1 lane)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc
d = int(sys.argv[1]); Ns = [int(a) for a in sys.argv[2:]]
G = 1000
for N in Ns:
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    row = []
    for lanes in (100, 16 if d == 20 else 8, 1):
        try:
            e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(d)],
                               eps_scale=w["eps_scale"], seed=1, target=w["target"], lanes_per_chain=lanes)
        except demc.DemczError:
            row.append("   n/a"); continue
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, w["gamma"]); e.synchronize()
        t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
        row.append(f"{dt / (G / 10) * 1e6:6.1f}")
        e.close()
    print(f"d={d} N={N:6d}  split / lanes / one lane: " + " / ".join(row) + " us per K-window", flush=True)
