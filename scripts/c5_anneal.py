"""C5 through the annealer's own driver (demcz_anneal: gamma adapted every 500 generations, demcz_anneal.jl:48-57):
wall time of the 10 000 generations, acceptance ratio per 1000, final gamma.
usage: python scripts/c5_anneal.py   (DEMCZ_NO_LR_SPEC=1: sixteen chains per workgroup, one generation per pass)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc

d, N, G = 10, 2048, 10000
w = demc.workloads.linreg_problem(d, N)
opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], γ=w["gamma"], verbose=False, T0=3, TN=1e-3, autostop="no")
for rep in range(2):
    t0 = time.perf_counter()
    mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], opts, seed=319531501)
    dt = time.perf_counter() - t0
    print(f"run {rep}: {dt:.3f} s end to end (history copied back) -> {N * G / dt:.3e} updates/s")
ch = (np.diff(mc.log_obj, axis=1) != 0)
print("acceptance per 1000 generations:", " ".join(f"{ch[:, a:a + 1000].mean():.3f}" for a in range(0, G - 1, 1000)))
