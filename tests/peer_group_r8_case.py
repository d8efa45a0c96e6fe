"""Eight replicas of 128 chains on ONE GPU, concurrently (tests/test_gpu_peer.py::test_eight_replicas_concurrently_on_one_gpu starts
this as a fresh process with GPU_MAX_HW_QUEUES=8 in its environment: the variable is read when the HIP runtime initialises).
usage: python tests/peer_group_r8_case.py <outdir> <d> <generations>"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    outdir, d, G = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import numpy as np
    import demc_jl_amd as demc
    R, N, K, seed = 8, 1024, 10, 808 + d
    n = N // R
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    es = []
    for r in range(R):
        e = demc.HipEngine(N=n, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                           target=w["target"], chain_id0=r * n)
        e.set_state(w["Zinit"][-N:][r * n:(r + 1) * n], None, w["Zinit"])
        es.append(e)
    demc.HipEngine.peer_group(es)
    half = (G // 2 // K) * K
    for a, b in ((1, half), (half + 1, G)):
        for e in es:
            e.run(a, b, w["gamma"])
    for e in es:
        e.synchronize()
    hist = [e.get_history(1, G) for e in es]
    sts = [e.get_state() for e in es]
    live = [e.live_status() for e in es]
    np.savez(outdir / "r8.npz", chain=np.concatenate([h[0] for h in hist], axis=0), log_obj=np.concatenate([h[1] for h in hist], axis=0),
             X=np.concatenate([s[0] for s in sts], axis=0), M=np.array([s[3] for s in sts]), Z0=np.array(sts[0][2]), Z7=np.array(sts[7][2]),
             live=np.array([int(l[0]) for l in live]), redos=np.array([l[1] for l in live]),
             launches=np.array([e.info()["window_launches"] for e in es]))
    for e in es:
        e.close()


if __name__ == "__main__":
    main()
