"""Diagnosis (round 5): a 300-slab demcz_run_checked whose hand-off fails in slab 280, against an undisturbed twin: where do the
R-hat trace / the history differ, if they do?  usage: python scripts/probes/redo_trace_diff.py [repeats]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import demc_jl_amd as demc

N, d, K, every, seed = 256, 5, 10, 4, 53
G = every * 300
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]


def one(fault, thr=0.0):
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], lanes_per_chain=164)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    if fault:
        e.debug_set_live_fault(1, every * 280)
        e.set_live_rearms(0)
    g_stop, trace, last = e.run_checked(1, G, w["gamma"], every, thr)
    ch, lo = e.get_history(1, g_stop)
    tr2 = np.array([np.nanmax(e.rhat(g - every + 1, g)) for g in range(every, g_stop + 1, every)])
    st = e.live_status()
    e.close()
    return trace, ch, lo, tr2, st


ref = one(False)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    a = one(True)
    bad = np.nonzero(a[0] != ref[0])[0]
    print(f"run {i}: live {a[4]}  trace differs at {bad[:10]} (of {len(a[0])});  history equal: {np.array_equal(a[1], ref[1])};  "
          f"trace recomputed after the call equal to the twin's: {np.array_equal(a[3], ref[3])}; twin's own monitor == recomputed: {np.array_equal(ref[0], ref[3])}", flush=True)
    for j in bad[:4]:
        print("   check", j, "monitor", a[0][j], "twin", ref[0][j], "recomputed", a[3][j])
