"""End-to-end wall time of demcz_sample on C2 including set-up, history download over PCIe and
host-side array assembly (the boundary hands results back in host memory)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc
d, N, G = 5, 1024, 10000
w = demc.workloads.mvnormal_problem(d, N)
opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="no")
demc.demcz_sample(w["target"], w["Zinit"], opts, seed=1)      # warm-up (module load, first-touch)
t0 = time.perf_counter()
mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=2)
dt = time.perf_counter() - t0
print(f"C2 end-to-end demcz_sample: {dt*1e3:.1f} ms -> {N*G/dt:.3e} chain-updates/s incl. PCIe download of {mc.chain.nbytes/1e6:.0f}+{mc.log_obj.nbytes/1e6:.0f} MB")
