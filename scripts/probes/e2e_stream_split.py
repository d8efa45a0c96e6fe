"""Where an end-to-end C2 demcz_sample goes with the streamed history: create + set_state, run (enqueue), history view (waits for
the last copies), state, close.  usage: python scripts/probes/e2e_stream_split.py"""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import demc_jl_amd as demc
d, N, G = 5, 1024, 10000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for rep in range(3):
    t0 = time.perf_counter()
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    t0b = time.perf_counter()
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.history_stream(True)
    t1 = time.perf_counter()
    e.run(1, G, w["gamma"])
    t2 = time.perf_counter()
    ch, lo = e.take_history(1, G)
    t3 = time.perf_counter()
    X, lp, Z, M = e.get_state()
    t4 = time.perf_counter()
    e.close()
    t5 = time.perf_counter()
    print(f"create {1e3*(t0b-t0):.1f} ms, set_state+stream on {1e3*(t1-t0b):.1f}, run (enqueue) {1e3*(t2-t1):.1f}, take_history {1e3*(t3-t2):.1f} "
          f"({(ch.nbytes+lo.nbytes)/1e9/(t3-t1):.1f} GB/s from run start), get_state {1e3*(t4-t3):.1f}, close {1e3*(t5-t4):.1f}; total {1e3*(t5-t0):.1f} ms", flush=True)
    del ch, lo
