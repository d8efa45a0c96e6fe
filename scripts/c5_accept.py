import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc
G = 2000
w = demc.workloads.linreg_problem(10, 2048)
N, d = 2048, 10
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(10)], eps_scale=w["eps_scale"], seed=31953150, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
temps = np.array([demc.tempbaseline(g, 2 * G, 3, 1e-3) for g in range(1, 2 * G + 1)])
e.run(1, G, w["gamma"], temps[:G]); e.run(G + 1, 2 * G, w["gamma"], temps[G:]); e.synchronize()
for a in range(1, 2 * G, 500):
    print(a, float(np.mean(e.accept_ratio(a, a + 499))))
