"""What a 1000-generation slab of C2 costs beyond its window kernel: wall time per slab of (a) demcz_run_checked (a check per slab: the
bench's step), (b) ONE demcz_run call over the same generations (the library cuts it into launches as long as a record buffer
holds: no check, no pacing between them), (c) one demcz_run call per slab, nothing waited for in between.  No event markers.
usage: python scripts/step_gap.py [slabs]        (DEMCZ_NO_HOST_PACING=1: (a) without the host waiting for the producer's event)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, d, K, every = 1024, 5, 10, 1000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for mode in ("checked", "one run call", "run call per slab", "checked", "one run call", "run call per slab"):
    G = (S + 5) * every
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run_checked(1, 5 * every, w["gamma"], every, 0.0)
    e.synchronize()
    t0 = time.perf_counter()
    if mode == "checked":
        e.run_checked(5 * every + 1, G, w["gamma"], every, 0.0)
    elif mode == "one run call":
        e.run(5 * every + 1, G, w["gamma"])
    else:
        for s in range(5, S + 5):
            e.run(s * every + 1, (s + 1) * every, w["gamma"])
    e.synchronize()
    dt = time.perf_counter() - t0
    launches = e.info()["window_launches"]
    print(f"{mode:18s}: {dt / S * 1e6:7.1f} us per 1000 generations (wall, {S} slabs); window launches so far {launches}", flush=True)
    e.close()
