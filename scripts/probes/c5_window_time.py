"""C5's generation loop alone: us per K-window.  usage: python scripts/probes/c5_window_time.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import demc_jl_amd as demc
N, d, G = 2048, 10, 2000
w = demc.workloads.linreg_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"]); e.run(1, 400, w["gamma"]); e.synchronize()
t = time.perf_counter(); e.run(401, G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t
print(f"C5: {dt / ((G - 400) / 10) * 1e6:.1f} us per K-window, launches {e.info()['window_launches']}")
e.close()
