"""Long bit-equality runs of the wave-per-chain consumers against the one-lane fused kernel (same spec, different code):
rare timing-dependent faults (a DMA landing where LDS reads are still queued, say) only show in millions of passes.
usage: python scripts/ps_stress.py [generations]   (DEMCZ_NO_LIVE=1: one launch per K-window)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ok = True
for d, N, K, tempered in ((5, 1024, 10, False), (5, 1000, 7, True), (3, 700, 10, False), (2, 513, 3, True), (4, 1024, 1000, False), (20, 1024, 10, False), (20, 600, 7, True)):
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    T = np.linspace(3.0, 0.5, G) if tempered else None
    res = {}
    for lanes in (164, 1):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=11 + d,
                           target=w["target"], lanes_per_chain=lanes)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        cuts = [0, G // 7, G // 2 + 3, G]
        for a, b in zip(cuts[:-1], cuts[1:]):
            e.run(a + 1, b, w["gamma"], None if T is None else T[a:b])
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        res[lanes] = (ch, lo, X, lp, Z, e.changed_total(1, G), e.live_status())
        e.close()
    a, b = res[164], res[1]
    same = all(np.array_equal(x, y) for x, y in zip(a[:5], b[:5])) and a[5] == b[5]
    ok &= same
    print(f"d={d:2d} N={N:5d} K={K:4d} tempered={tempered!s:5s} generations={G}: {'identical' if same else 'DIFFERENT'}  (LIVE on: {a[6][0]}, redos {a[6][1]})", flush=True)
sys.exit(0 if ok else 1)
