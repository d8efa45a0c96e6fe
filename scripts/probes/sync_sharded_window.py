"""What one K-window of a SHARDED run costs with a communicator of one rank (the only kind a one-GPU box can make): the
synchronous schedule (append_lag 0: window kernel -> all-gather -> scatter per K generations) and deferred batches.
usage: python scripts/probes/sync_sharded_window.py [N] [d]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import demc_jl_amd as demc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K, G = 10, 4000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for lag in (0, 10, 25, 50):
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    e.comm_init(e.comm_unique_id(), 1, 0)
    if lag:
        e.set_append_lag(lag)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, 1000, w["gamma"]); e.synchronize()
    t = time.perf_counter(); e.run(1001, G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t
    nl = e.info()["window_launches"]
    print(f"N={N} d={d} append_lag={lag:2d}: {dt / ((G - 1000) / K) * 1e6:6.1f} us per K-window = {N * (G - 1000) / dt:.3e} updates/s  ({nl} window launches)", flush=True)
    e.close()
