"""Regression target at C5's size: the two-generations-per-pass kernel (window_kernel_lr8s, producer kernel beside it) against
the one-generation kernel (window_kernel_lr16), bit for bit, over a long annealed run -- timing-dependent faults only show at
full size and over many launches.   usage: python scripts/lr_check.py [gens] [gamma] [repeats] [sync]     (sync: a demcz_synchronize between the two calls)"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
gam = float(sys.argv[2]) if len(sys.argv) > 2 else 2.38
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SYNC = len(sys.argv) > 4
N, d = 2048, 10
w = demc.workloads.linreg_problem(d, N)
M0 = w["Zinit"].shape[0]
temps = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])


def run(spec, timing):
    if spec:
        os.environ.pop("DEMCZ_NO_LR_SPEC", None)
    else:
        os.environ["DEMCZ_NO_LR_SPEC"] = "1"
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=31953150, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    h = G // 2
    e.run(1, h, gam, temps[:h])
    if SYNC:
        e.synchronize()
    if timing:
        e.set_kernel_timing(True)
    e.run(h + 1, G, gam, temps[h:])
    e.synchronize()
    acc_dev = float(np.mean(e.accept_ratio(h + 1, G)))
    lo = e.get_history(1, G)[1]
    acc_hist = float((np.diff(lo[:, h:], axis=1) != 0).mean())
    if abs(acc_dev - acc_hist) > 1e-12:
        print("   device acceptance ratio", acc_dev, "history", acc_hist)
    X, lp, Z, M = e.get_state()
    st = e.live_status()
    e.close()
    return lo, X, Z, st


first = run(True, True)                 # (first thing on the device: code objects load, clocks ramp)
ref = run(False, False)
print("reference (one generation per pass): live", ref[3], "acceptance", float((np.diff(ref[0], axis=1) != 0).mean()), "second half", float((np.diff(ref[0][:, G // 2:], axis=1) != 0).mean()))
print("the very first run identical:", np.array_equal(first[0], ref[0]) and np.array_equal(first[2], ref[2]), "second-half acceptance", float((np.diff(first[0][:, G // 2:], axis=1) != 0).mean()))
for r in range(reps):
    for timing in (False, True):
        got = run(True, timing)
        same = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        first = None
        if not same:
            bad = np.argwhere(got[0] != ref[0])
            first = (int(bad[:, 1].min()), int(len(bad)))
        print(f"two generations per pass, repeat {r}, kernel timing {timing}: live {got[3]} identical {same}" + (f"  first differing generation {first[0]} ({first[1]} entries)" if first else ""), flush=True)
