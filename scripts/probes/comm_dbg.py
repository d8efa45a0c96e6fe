import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import demc_jl_amd as demc
from demc_jl_amd._lib import DemczError
def eng(lag, G=400, N=256, d=5, K=10, seed=5):
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed, target=w["target"])
    e.comm_init(e.comm_unique_id(), 1, 0)
    if lag: e.set_append_lag(lag)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    return e, w
for rep in range(3):
  for entry in ("run", "checked"):
    e, w = eng(2)
    e.run(1, 40, w["gamma"]); e.synchronize()
    e.set_comm_timeout(50); e.debug_stall_exchange(1500)
    t0 = time.perf_counter()
    try:
        if entry == "checked": e.run_checked(41, 400, w["gamma"], 40, 0.0)
        else:
            e.run(41, 400, w["gamma"]); e.synchronize()
        print(entry, "NO RAISE", time.perf_counter() - t0, flush=True)
    except DemczError as ex:
        print(entry, "raised", ex.code, round(time.perf_counter() - t0, 3), flush=True)
    t1 = time.perf_counter(); e.close(); print("  close", round(time.perf_counter() - t1, 3), flush=True)
