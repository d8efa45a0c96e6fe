"""Wall time per 1000-generation slab of demcz_run_checked (C2, monitor mode) with and without the per-call kernel-timing
events: what the stream markers between two window launches cost.  usage: python scripts/step_overhead.py [slabs]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, d, K, every = 1024, 5, 10, 1000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for timing in (False, True, False, True):
    G = (S + 5) * every
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run_checked(1, 5 * every, w["gamma"], every, 0.0)
    e.synchronize()
    if timing:
        e.set_kernel_timing(True)
    t0 = time.perf_counter()
    e.run_checked(5 * every + 1, G, w["gamma"], every, 0.0)
    e.synchronize()
    dt = time.perf_counter() - t0
    extra = ""
    if timing:
        n, ms = e.get_kernel_time()
        extra = f"  window kernels {ms * 1e3 / max(n, 1):.1f} us per launch ({n} launches)"
    e.close()
    print(f"kernel timing {'on ' if timing else 'off'}: {dt / S * 1e6:7.1f} us per slab (wall){extra}", flush=True)
