"""One comm-failure scenario in a process of its own (tests/test_gpu_comm_failure.py starts it): after ncclCommAbort nothing about
the process's RCCL state is promised -- "a fresh process is the only retry" -- so every abort scenario gets a fresh process.
usage: comm_failure_case.py <lag> <run+synchronize|run_checked>; prints one JSON line."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import demc_jl_amd as demc
from demc_jl_amd._lib import DemczError


def main(lag, entry):
    G, N, d, K, seed = 400, 256, 5, 10, 5
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.comm_init(e.comm_unique_id(), 1, 0)          # a communicator of one rank: every RCCL call of the data path runs
    if lag:
        e.set_append_lag(lag)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, 40, w["gamma"])
    e.synchronize()                                # healthy so far
    e.set_comm_timeout(50)
    e.debug_stall_exchange(1500)                   # the next collective is held back for 1.5 s: far beyond the deadline
    out = {"lag": lag, "entry": entry, "raised": None, "later": []}
    t0 = time.perf_counter()
    try:
        if entry == "run_checked":
            e.run_checked(41, 400, w["gamma"], 40, 0.0)
        else:
            e.run(41, 400, w["gamma"])
            e.synchronize()
    except DemczError as ex:
        out["raised"] = ex.code
        out["message"] = str(ex)
    out["seconds_to_surface"] = time.perf_counter() - t0
    for name, call in (("run", lambda: e.run(401, 402, w["gamma"])), ("synchronize", e.synchronize),
                       ("get_history", lambda: e.get_history(1, 10)), ("get_state", e.get_state)):
        try:                                       # the handle is dead: every call says so at once
            call()
            out["later"].append([name, None])
        except DemczError as ex:
            out["later"].append([name, ex.code])
    t1 = time.perf_counter()
    e.close()                                      # and destroying it does not hang either
    out["seconds_to_close"] = time.perf_counter() - t1
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]), sys.argv[2])
