"""One rank of a TWO-PROCESS run on ONE GPU whose K-boundary rows are handed over inside the launches through HIP IPC
(demcz_peer_export / demcz_peer_attach; tests/test_gpu_peer.py spawns the ranks): each process holds its own archive replica in
fine-grained device memory, opens the other's over IPC, and its publisher waves store a boundary's rows into both.  The host side
(this script) carries the 64-byte handles and makes the ranks meet, over torch.distributed / gloo.
usage: python tests/peer_ipc_case.py <rank> <world> <port> <outdir> <d> <generations, comma separated pieces>"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    rank, world, port, outdir, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4]), int(sys.argv[5])
    pieces = [int(v) for v in sys.argv[6].split(",")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import demc_jl_amd as demc
    N, K, seed = 1024, 10, 2024 + d
    G = sum(pieces)
    n = N // world
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=n, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], chain_id0=rank * n)
    e.set_state(w["Zinit"][-N:][rank * n:(rank + 1) * n], None, w["Zinit"])
    handles = [None] * world
    dist.all_gather_object(handles, e.peer_export(world, rank))
    e.peer_attach(handles)
    status = e.peer_status()
    dist.barrier()                          # every replica is filled and opened before anybody publishes
    g = 1
    for p in pieces:
        e.run(g, g + p - 1, w["gamma"])
        g += p
    e.synchronize()
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    live, info = e.live_status(), e.info()
    dist.barrier()                          # nobody closes a mapping while another rank may still be publishing through its own
    e.peer_detach()                         # this rank's mappings of the other archives are closed ...
    dist.barrier()                          # ... on every rank, before any rank frees its exported archive (demcz_destroy)
    np.savez(outdir / f"rank{rank}.npz", chain=ch, log_obj=lo, X=X, logp=lp, Z=np.array(Z), M=M, mode=status[0], peers=status[1],
             live=int(live[0]), redos=live[1], launches=info["window_launches"])
    e.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
