"""What one K-window of the SAMPLER-PRESERVING sharded schedule (rows visible from the next generation on, append_lag 0) costs on
the one GPU this box has, per way of getting a boundary's rows into every replica:

  exchange   demcz_comm_init with a one-rank communicator, DEMCZ_NO_PEER: window kernel -> ncclAllGather -> scatter per K-window
             (the only schedule there was until round 4)
  ipc-self   the same communicator with the IPC set-up (DEMCZ_PEER_SELF): fine-grained archive, LIVE launches through the
             boundaries, error words max-reduced at verification -- what a rank of a multi-GPU run executes, minus the peers
  group R    R handles of this process on the one device, each with its own replica and N / R chains, publishing into each
             other's replicas from inside their launches (demcz_peer_group): the full protocol, the stores to "peers" staying
             on the device instead of crossing xGMI
  one handle the unsharded LIVE run of all N chains, for reference

usage: python scripts/probes/sync_sharded_window.py [N_total] [d] [generations]"""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import demc_jl_amd as demc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = int(sys.argv[2]) if len(sys.argv) > 2 else 5
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
K, WARM = 10, 1000
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]


def engines(R, comm=None):
    n = N // R
    es = []
    for r in range(R):
        e = demc.HipEngine(N=n, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                           target=w["target"], chain_id0=r * n)
        e.set_state(w["Zinit"][-N:][r * n:(r + 1) * n], None, w["Zinit"])
        es.append(e)
    if comm is not None:
        os.environ.pop("DEMCZ_NO_PEER", None)
        os.environ.pop("DEMCZ_PEER_SELF", None)
        os.environ[comm] = "1"
        es[0].comm_init(es[0].comm_unique_id(), 1, 0)
    elif R > 1:
        demc.HipEngine.peer_group(es)
    return es


def timed(label, es):
    for e in es:
        e.run(1, WARM, w["gamma"])
    for e in es:
        e.synchronize()
    t = time.perf_counter()
    for e in es:
        e.run(WARM + 1, G, w["gamma"])
    for e in es:
        e.synchronize()
    dt = time.perf_counter() - t
    nl = [e.info()["window_launches"] for e in es]
    live = [e.live_status() for e in es]
    print(f"N={N} d={d} {label:12s}: {dt / ((G - WARM) / K) * 1e6:6.2f} us per K-window = {N * (G - WARM) / dt:.3e} updates/s  "
          f"(window launches per handle {nl[0]}, LIVE {live[0]}, peer status {es[0].peer_status()}, kernel {es[0].kernel_name()})", flush=True)
    for e in es:
        e.close()


timed("one handle", engines(1))
timed("exchange", engines(1, "DEMCZ_NO_PEER"))
timed("ipc-self", engines(1, "DEMCZ_PEER_SELF"))
for R in (2, 4, 8):
    if N % R == 0:
        timed(f"group R={R}", engines(R))
