import sys, time
sys.path.insert(0, '/root/repo')
import demc_jl_amd as demc
N, d, G = 4096, 20, 2000
w = demc.workloads.mvnormal_problem(d, N); blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"]); e.run(1, 400, w["gamma"]); e.synchronize()
t = time.perf_counter(); e.run(401, G, w["gamma"]); e.synchronize(); dt = time.perf_counter() - t
print(f"C3: {dt / ((G - 400) / 10) * 1e6:.1f} us per K-window, launches {e.info()['window_launches']}")
e.close()
