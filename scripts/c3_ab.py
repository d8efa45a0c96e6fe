"""C3 (MvNormal d=20 in four blocks of five, N=4096): us per K-window of the block kernel's forms.
DEMCZ_NO_MLB_INCREMENTAL=1: full evaluation in the grouped order (group-start mask) instead of the incremental form.
usage: python scripts/c3_ab.py [gens] [swapped]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N, d, K = 4096, 20, 10
w = demc.workloads.mvnormal_problem(d, N)
blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
if len(sys.argv) > 2 and sys.argv[2] == "swapped":     # the same work with the last two blocks in the other order: the sums are then NOT
    blocks = [range(0, 5), range(5, 10), range(15, 20), range(10, 15)]      # cut at the block boundaries -> rounds 1-3's kernel and order
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=blocks, eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G, w["gamma"]); e.synchronize()
e.set_kernel_timing(True)
t0 = time.perf_counter(); e.run(G + 1, 2 * G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t0
n, ms = e.get_kernel_time()
print(f"C3 {e.kernel_name()}: {dt / (G / K) * 1e6:.2f} us per K-window wall, {ms * 1e3 / (G / K):.2f} us in window kernels ({n} launches), live {e.live_status()}", flush=True)
e.close()
