"""The sharded data path on ONE GPU (demcz_comm_init with nranks = 1): what a batch of the deferred schedule costs end to end --
window launch, snapshot, ncclAllGather on the side stream's communicator, scatter kernel, the events between them -- against
the same schedule without a communicator.  Wall time per generation for E = 10, 25, 50 boundaries per batch.
usage: python scripts/rccl_single_rank_lag.py [gens] [d] [checked]     (d = 20: C4's per-GPU shard; checked: through
demcz_run_checked with a check every 1000 generations, as bench.py runs it)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N, K = 1024, 10
d = int(sys.argv[2]) if len(sys.argv) > 2 else 5
CHECKED = len(sys.argv) > 3
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
for comm in (False, True):
    for E in (10, 25, 50):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * G // K + 1), Gcap=2 * G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=3, target=w["target"])
        if comm:
            e.comm_init(e.comm_unique_id(), 1, 0)
        e.set_append_lag(E)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        if CHECKED:
            e.run_checked(1, G, w["gamma"], 1000, 0.0)     # (the first checked call makes the check's buffers)
        else:
            e.run(1, G, w["gamma"])
        e.synchronize()
        e.set_kernel_timing(True)
        t0 = time.perf_counter()
        if CHECKED:
            e.run_checked(G + 1, 2 * G, w["gamma"], 1000, 0.0)
        else:
            e.run(G + 1, 2 * G, w["gamma"])
        t_enq = time.perf_counter() - t0        # the call returns when everything is enqueued
        e.synchronize()
        dt = time.perf_counter() - t0
        n, ms = e.get_kernel_time()
        print(f"{'RCCL nranks=1' if comm else 'no communicator'}  E={E:2d}: {dt / (G / (K * E)) * 1e6:7.1f} us wall per batch ({t_enq / (G / (K * E)) * 1e6:5.1f} us of host time to enqueue it), window kernels {ms * 1e3 / max(n, 1):6.1f} us per launch "
              f"({n} launches) -> {N * G / dt:.3e} updates/s", flush=True)
        e.close()
