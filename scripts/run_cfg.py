"""Run one BASELINE config for profiling: python scripts/run_cfg.py C5|C3|C2|C4 [gens]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc
name = sys.argv[1]; G = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if name == "C5": N, d = 2048, 10; w = demc.workloads.linreg_problem(d, N); blocks = [range(d)]
elif name == "C3": N, d = 4096, 20; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
elif name == "C4": N, d = 1024, 20; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(d)]
else: N, d = 1024, 5; w = demc.workloads.mvnormal_problem(d, N); blocks = [range(d)]
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"]); e.run(1, G, w["gamma"]); e.synchronize(); e.close()
