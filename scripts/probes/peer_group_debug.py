"""Replica group at d = 20 (window_kernel_pw): does the in-launch hand-off hold, and how long do the calls take?
usage: python scripts/probes/peer_group_debug.py [R] [d] [pieces, comma separated] [repeats]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import demc_jl_amd as demc

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pieces = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [250, 350]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N, K = 1024, 10
G = sum(pieces)
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for rep in range(reps):
    n = N // R
    es = []
    for r in range(R):
        e = demc.HipEngine(N=n, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=5 + rep,
                           target=w["target"], chain_id0=r * n)
        e.set_state(w["Zinit"][-N:][r * n:(r + 1) * n], None, w["Zinit"])
        es.append(e)
    if R > 1:
        demc.HipEngine.peer_group(es)
    g = 1
    t0 = time.perf_counter()
    for p in pieces:
        for e in es:
            e.run(g, g + p - 1, w["gamma"])
        g += p
    t1 = time.perf_counter()
    for e in es:
        e.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: R={R} d={d} pieces={pieces}: enqueue {1e3 * (t1 - t0):.2f} ms, drain {1e3 * (t2 - t1):.2f} ms, live {[e.live_status() for e in es]}, "
          f"launches {[e.info()['window_launches'] for e in es]}, kernel {es[0].kernel_name()}", flush=True)
    for e in es:
        e.close()
