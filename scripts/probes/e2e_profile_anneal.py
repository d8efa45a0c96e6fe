"""cProfile of demcz_anneal on C5 (host-side time around the library calls)."""
import cProfile, pstats, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import demc_jl_amd as demc
d, N, G = 10, 2048, 10000
w = demc.workloads.linreg_problem(d, N)
opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], γ=w["gamma"], verbose=False, T0=3, TN=1e-3, autostop="no")
demc.demcz_anneal(w["target"], w["Zinit"], opts, seed=1)
pr = cProfile.Profile(); pr.enable()
mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], opts, seed=2)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
