"""C5 (regression target, d = 10, nobs = 1000, N = 2048, annealed): window-kernel time per launch, launches, LIVE state.
usage: python scripts/c5_steps.py [gens] [gamma]   (gamma: default the workload's 2.0 -> ~1 % acceptance; ~0.5 gives the 0.2-0.4 the annealer's adaptation steers for)   (DEMCZ_NO_LR_SPEC=1: sixteen chains per workgroup, one generation per pass)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
GAMMA = float(sys.argv[2]) if len(sys.argv) > 2 else None
N, d = 2048, 10
w = demc.workloads.linreg_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                   seed=31953150, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
gam = GAMMA if GAMMA is not None else w["gamma"]
temps = np.array([demc.tempbaseline(g, 2 * G, 3, 1e-3) for g in range(1, 2 * G + 1)])
e.run(1, G, gam, temps[:G]); e.synchronize()
e.set_kernel_timing(True)
t0 = time.perf_counter()
e.run(G + 1, 2 * G, gam, temps[G:]); e.synchronize()
dt = time.perf_counter() - t0
n, ms = e.get_kernel_time()
acc = float(np.mean(e.accept_ratio(G + 1, 2 * G)))
print(f"gamma={gam} gens={G} wall={dt*1e3:.2f} ms  window kernels: {n} launches, {ms:.3f} ms -> {ms*1e3/(G/10):.2f} us per K-window; "
      f"live={e.live_status()} accept={acc:.4f} updates/s={N*G/dt:.3e}")
