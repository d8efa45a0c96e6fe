"""Convergence figures of BASELINE C3 / C4 (d=20) over a long run: mean error in sd units, R-hat, accept band.
usage: python scripts/c3_stats.py [gens]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for name, N, blocks in (("C3", 4096, [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]), ("C4-shard", 1024, [range(20)])):
    d = 20
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10), Gcap=1000, blockindex=blocks, eps_scale=w["eps_scale"],
                       seed=31953150, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    sd = np.sqrt(np.diag(w["Sigma"]))
    for g0 in range(0, G, 1000):
        e.synchronize(); e.set_history_origin(g0)
        e.run(g0 + 1, g0 + 1000, w["gamma"])
        mean, cov = e.mean_cov(g0 + 1, g0 + 1000)
        rh = e.rhat(g0 + 1, g0 + 1000)
        acc = e.accept_ratio(g0 + 1, g0 + 1000)
        print(f"{name} gens {g0+1:6d}-{g0+1000:6d}: max|mean-mu|/sd={np.abs((mean-w['mu'])/sd).max():6.3f}  "
              f"var ratio={np.diag(cov).sum()/np.diag(w['Sigma']).sum():6.3f}  maxRhat={rh.max():6.3f}  "
              f"accept [{acc.min():.3f}, {acc.max():.3f}]", flush=True)
    e.close()
