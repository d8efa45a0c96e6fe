"""Block updates at C3's size (d = 20, four blocks of five, N = 4096): the split block kernel (four-wave workgroups, LIVE launches)
against the fused 16-lane block kernel, bit for bit, over a long run.  usage: python scripts/mlb_check.py [gens] [repeats]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, d = 4096, 20
w = demc.workloads.mvnormal_problem(d, N)
blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
M0 = w["Zinit"].shape[0]


def run(lanes):
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=5, target=w["target"],
                       lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G // 3, w["gamma"])
    e.run(G // 3 + 1, G, w["gamma"])
    e.synchronize()
    lo = e.get_history(1, G)[1]
    X, lp, Z, M = e.get_state()
    info = (e.info()["lanes_per_chain"], e.live_status())
    e.close()
    return lo, X, Z, info


ref = run(16)
print("fused 16-lane block kernel:", ref[3])
for r in range(reps):
    got = run(0)
    same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    print(f"split block kernel, repeat {r}: {got[3]} identical {same}", flush=True)
