"""Regression target (C5 shape): us per generation against the number of observations -- separates the per-generation
overhead (draw records, gather, barrier, tree, history) from the matrix-core work.  usage: lr_probe.py [N] [lanes]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d, G = 10, 1000
for nobs in (64, 256, 1000, 1984):
    w = demc.workloads.linreg_problem(d, N, nobs=nobs)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                       target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    T = np.array([demc.tempbaseline(g, 2 * G, 3, 1e-3) for g in range(1, 2 * G + 1)])
    e.run(1, G, w["gamma"], T[:G]); e.synchronize()
    t0 = time.perf_counter()
    e.run(G + 1, 2 * G, w["gamma"], T[G:]); e.synchronize()
    dt = time.perf_counter() - t0
    flop = 2 * nobs * d + 3 * nobs
    print(f"N={N} lanes={e.info()['lanes_per_chain']} nobs={nobs:5d}: {dt / G * 1e6:7.3f} us per generation, {N * G / dt:9.3e} upd/s, "
          f"{N * G / dt * flop / 1e12:6.2f} TFLOP/s algorithmic", flush=True)
    e.close()
