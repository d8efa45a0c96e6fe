"""One C2-shaped LIVE run on window_kernel_ps3 (DEMCZ_PS3=1, read once when the library is loaded -- hence a process of its
own, started by tests/test_gpu_live.py) compared with the CPU oracle bit for bit.  Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

os.environ["DEMCZ_PS3"] = "1"
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

import demc_jl_amd as demc
import oracle_py as oracle
from helpers import oracle_sample

N, d, K, G, seed = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 5, 10, 1500, 77
oracle.lib()
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, 500, w["gamma"])
e.run(501, 1500, w["gamma"])
ch, lo = e.get_history(1, G)
X, lp, Z, M = e.get_state()
counts, live = e.kernel_counts(), e.live_status()
e.close()
ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=max(1, min(len(os.sched_getaffinity(0)), 8)))
same = bool(np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"]) and np.array_equal(X, ref["X"]) and
            np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"]))
print(json.dumps({"same": same, "counts": counts, "live": list(live)}))
