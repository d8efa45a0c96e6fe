"""The reference's regression example at its own dimension (test/example_linreg.jl:9: d = 26, nobs = 1000), N = 1024, annealed: the
library's choice (window_kernel_ml<LINREG_SSE, 26, 16>, round 5) against the one-lane kernel it used to fall to.
usage: python scripts/linreg_d26.py [generations] [nobs] [lanes ...]      (DEMCZ_NO_ML_COOP=1: without the helper waves)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, d, K = 1024, 26, 10
NOBS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
w = demc.workloads.linreg_problem(d, N, nobs=NOBS)
M0 = w["Zinit"].shape[0]
T = np.array([demc.tempbaseline(g, 2 * G, 3.0, 1e-3) for g in range(1, 2 * G + 1)])
for lanes in ([int(a) for a in sys.argv[3:]] or [0, 1]):
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                       target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G, w["gamma"], T[:G]); e.synchronize()
    t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"], T[G:]); e.synchronize(); dt = time.perf_counter() - t0
    print(f"nobs={NOBS} lanes_per_chain={lanes}: {e.kernel_name()}: {dt / (G / K) * 1e6:.1f} us per K-window = {N * G / dt:.3e} updates/s", flush=True)
    e.close()
