"""Chain-updates/s of every BASELINE config that fits one GPU (C2, C3, C4's per-GPU shard, C5) plus
the CPU rows of BASELINE.md section 2.  Prints one line per row; run on the GPU box.
usage: python scripts/bench_configs.py [gens]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def gpu_row(name, w, N, d, blocks, anneal=False):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (2 * G // 10 + 1), Gcap=2 * G, blockindex=blocks, eps_scale=w["eps_scale"],
                       seed=31953150, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    temps = np.array([demc.tempbaseline(g, 2 * G, 3, 1e-3) for g in range(1, 2 * G + 1)]) if anneal else None
    e.run(1, G, w["gamma"], None if temps is None else temps[:G]); e.synchronize()
    t0 = time.perf_counter()
    e.run(G + 1, 2 * G, w["gamma"], None if temps is None else temps[G:]); e.synchronize()
    dt = time.perf_counter() - t0
    rh = e.rhat(G + 1, 2 * G)
    info = e.info()
    B = 8 * (3 * d + 1 + d / 10)
    print(f"{name:34s} GPU  N={N:6d} d={d:2d} lanes={info['lanes_per_chain']:2d}  {N*G/dt:10.3e} upd/s  window={dt/(G/10)*1e6:8.1f} us  "
          f"{N*G/dt*B/1e9:7.1f} GB/s alg ({N*G/dt*B/8e12*100:5.2f}% of 8 TB/s)  maxRhat={rh.max():.3f}", flush=True)
    e.close()


def cpu_rows():
    import oracle_py as O
    O.build(native=True)
    d, N, Gc = 5, 1024, 2000
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    for label, sched in (("CPU-fair 1 thread, synchronous", 0), ("CPU-fair 1 thread, sequential (reference order)", 1)):
        Mcap = M0 + N * Gc // 10
        prob = O.Problem(N, d, 10, Mcap, w["eps_scale"], 1, target=w["target"].spec())
        X = np.array(w["Zinit"][-N:], order="F"); lp = O.logp(prob, X)
        Z = np.zeros((Mcap, d), order="F"); Z[:M0] = w["Zinit"]
        t0 = time.perf_counter(); O.run(prob, X, lp, Z, M0, 1, Gc, 2.38, schedule=sched, native=True); dt = time.perf_counter() - t0
        print(f"{label:50s} C2 {N*Gc/dt:10.3e} upd/s", flush=True)
    ncores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    for thr in sorted({min(ncores, 4), min(ncores, 8)}):
        Mcap = M0 + N * Gc // 10
        prob = O.Problem(N, d, 10, Mcap, w["eps_scale"], 1, target=w["target"].spec())
        X = np.array(w["Zinit"][-N:], order="F"); lp = O.logp(prob, X)
        Z = np.zeros((Mcap, d), order="F"); Z[:M0] = w["Zinit"]
        t0 = time.perf_counter(); O.run(prob, X, lp, Z, M0, 1, Gc, 2.38, native=True, threads=thr); dt = time.perf_counter() - t0
        print(f"{'CPU-omp synchronous, %d threads (of %d cores)' % (thr, ncores):50s} C2 {N*Gc/dt:10.3e} upd/s", flush=True)
    # reference's O(M) index draw (collect(1:M) + deleteat!, demcz.jl:176-178): cost per block-step at archive size M
    L = O.lib(native=True)
    for M in (10**3, 10**5, 10**6):
        n = max(3, int(2e8 // M)); t0 = time.perf_counter()
        for i in range(n):
            L.oracle_faithful_index_cost(M, (i * 7919) % M)
        per = (time.perf_counter() - t0) / n
        print(f"CPU-faithful index-draw emulation at M={M:8d}: {per*1e6:9.1f} us per block-step -> <= {1/per:10.3e} upd/s", flush=True)


if __name__ == "__main__":
    w = demc.workloads.mvnormal_problem(5, 1024); gpu_row("C2 MvNormal d=5 N=1024", w, 1024, 5, [range(5)])
    w = demc.workloads.mvnormal_problem(20, 4096); gpu_row("C3 MvNormal d=20 4 blocks N=4096", w, 4096, 20, [range(0, 5), range(5, 10), range(10, 15), range(15, 20)])
    w = demc.workloads.mvnormal_problem(20, 1024); gpu_row("C4 shard MvNormal d=20 N=1024/GPU", w, 1024, 20, [range(20)])
    w = demc.workloads.mvnormal_problem(20, 8192); gpu_row("C4 whole MvNormal d=20 N=8192 on 1 GPU", w, 8192, 20, [range(20)])
    w = demc.workloads.linreg_problem(10, 2048); gpu_row("C5 linreg SSE d=10 nobs=1000 N=2048 anneal", w, 2048, 10, [range(10)], anneal=True)
    if len(sys.argv) <= 2: cpu_rows()
