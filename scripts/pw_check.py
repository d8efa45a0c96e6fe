import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
d, N, G, K = 20, int(sys.argv[1]), int(sys.argv[2]), 10
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
out = {}
for lanes in (164, 16):
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=7, target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    for a, b in ((1, G // 3), (G // 3 + 1, G)):
        e.run(a, b, w["gamma"])
    ch, lo = e.get_history(1, G)
    out[lanes] = (ch, lo, e.live_status())
    e.close()
a, b = out[164], out[16]
print("live", a[2], b[2], "chain equal", np.array_equal(a[0], b[0]), "logobj equal", np.array_equal(a[1], b[1]))
if not np.array_equal(a[1], b[1]):
    bad = np.argwhere(a[1] != b[1])
    print("first mismatches (chain, gen):", bad[np.argsort(bad[:, 1])][:8].tolist())
    c, g = bad[np.argsort(bad[:, 1])][0]
    print(a[1][c, g], b[1][c, g], a[1][c, max(g-1,0)], b[1][c, max(g-1,0)])
