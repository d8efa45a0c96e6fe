import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import demc_jl_amd as demc
d = 20
N = int(sys.argv[1]); G = int(sys.argv[2]); S = int(sys.argv[3])
w = demc.workloads.mvnormal_problem(d, N)
opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False)
a, Za, ra = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=31953150, return_runner=True)
sh = demc.Sharding(rank=0, world_size=1, mode="host", local_shards=S)
b, Zb, rb = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=31953150, sharding=sh, return_runner=True)
print("layouts", ra.engines[0].info()["lanes_per_chain"], rb.engines[0].info()["lanes_per_chain"])
print("chain equal", np.array_equal(a.chain, b.chain), "logobj equal", np.array_equal(a.log_obj, b.log_obj), "Z equal", np.array_equal(Za, Zb))
if not np.array_equal(a.log_obj, b.log_obj):
    bad = np.argwhere(a.log_obj != b.log_obj)
    bad = bad[np.argsort(bad[:, 1], kind="stable")]
    print("mismatching (chain, gen) count", len(bad), "first:", bad[:10].tolist())
    c, g = bad[0]
    print("values", a.log_obj[c, g], b.log_obj[c, g], "prev", a.log_obj[c, g - 1], b.log_obj[c, g - 1])
    print("chains mismatching at that gen:", np.unique(bad[bad[:, 1] == g][:, 0])[:20])
ra.close(); rb.close()
