#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference holds no fixtures for this path and cannot
be executed here (no Julia), so these vectors pin the arithmetic SPEC: any
change to the oracle or the kernels that moves a bit shows up against them.  Inputs are stored
next to the expected outputs, so the GPU tests need nothing but the .npz files."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import demc_jl_amd as demc      # noqa: E402
import oracle_py as O           # noqa: E402
from helpers import oracle_sample   # noqa: E402

OUT = Path(__file__).resolve().parent


def traj(name, w, N, G, blocks, seed, gamma, temperature=None, schedule=0, K=10):
    r = oracle_sample(O, w["target"], w["Zinit"], N, K, G, blocks, w["eps_scale"], gamma, seed,
                      temperature=temperature, schedule=schedule)
    spec = w["target"].spec()
    tgt = {f"target_{k}": v for k, v in spec.items() if k != "kind"}
    nb = [list(b) for b in (blocks or [range(w["d"])])]
    np.savez_compressed(OUT / f"{name}.npz", kind=spec["kind"], Zinit=w["Zinit"], eps_scale=w["eps_scale"], N=N, K=K, G=G,
                        seed=seed, gamma=gamma, schedule=schedule, block_offsets=np.cumsum([0] + [len(b) for b in nb]),
                        block_indices=np.concatenate(nb), temperature=(temperature if temperature is not None else np.zeros(0)),
                        chain=r["chain"], log_obj=r["log_obj"], Z=r["Z"], changed=r["changed"], **tgt)


def main():
    # (ii) short trajectories
    w = demc.workloads.mvnormal_problem(5, 4)
    traj("traj_mvn_d5_N4_sync", w, 4, 50, None, 31953150, 2.38, schedule=0)
    traj("traj_mvn_d5_N4_seq", w, 4, 50, None, 31953150, 2.38, schedule=1)
    w = demc.workloads.mvnormal_problem(6, 12)
    traj("traj_mvn_d6_blocks", w, 12, 30, [[0], [1, 2], [5, 3, 4]], 7, 2.38)
    w = demc.workloads.mvnormal_problem(20, 8)
    traj("traj_mvn_d20_blocks4", w, 8, 20, [range(0, 5), range(5, 10), range(10, 15), range(15, 20)], 9, 2.38)
    w = demc.workloads.iso_quad_problem(10, 6)
    T = np.array([2.0 * (1e-4 / 2.0) ** (g / 40) for g in range(1, 41)])
    traj("traj_isoquad_d10_anneal", w, 6, 40, None, 11, 2.38, temperature=T)
    w = demc.workloads.linreg_problem(10, 6, nobs=50)
    T = np.array([3.0 * (1e-3 / 3.0) ** (g / 30) for g in range(1, 31)])
    traj("traj_linreg_d10_anneal", w, 6, 30, None, 13, 2.0, temperature=T)

    # (i) single block-steps with every intermediate
    w = demc.workloads.mvnormal_problem(5, 8)
    prob = O.Problem(8, 5, 10, 80, w["eps_scale"], 31953150, blocks=[[0, 1, 2], [3], [4, 2]], target=w["target"].spec())
    Z = np.zeros((80, 5), order="F")
    Z[:50] = w["Zinit"][:50]
    rows = []
    rng = np.random.default_rng(0)
    for c, g, ib in [(0, 1, 0), (3, 1, 1), (7, 2, 2), (5, 1000, 0), (2, 123456789, 1)]:
        x = w["mu"] + 0.05 * rng.standard_normal(5)
        lp = float(O.logp(prob, x[None, :])[0])
        o = O.block_step(prob, Z, 50, c, g, ib, 2.38, x, lp)
        rows.append(dict(c=c, g=g, ib=ib, x0=x, lp0=lp, **o))
    np.savez_compressed(OUT / "block_steps.npz", Z=Z, M=50, seed=31953150, eps_scale=w["eps_scale"], mu=w["target"].mu,
                        W=w["target"].W, c0=w["target"].c0,
                        **{f"{k}_{i}": np.asarray(v) for i, r in enumerate(rows) for k, v in r.items()})

    # (iii) R-hat on a fixed tensor, (iv) log-density tables, draw pipeline
    rng = np.random.default_rng(42)
    chain = np.asfortranarray(np.cumsum(rng.standard_normal((6, 3, 41)), axis=2) * 0.05 + rng.standard_normal((6, 3, 1)))
    X = w["mu"] + 0.1 * rng.standard_normal((16, 5))
    words, normals, logu = [], [], []
    for blk in range(64):
        r1, r2 = O.draw_block(31953150, 5, 1000 + blk)
        words.append((r1, r2))
        normals.append(O.normal_pair(r1, r2))
        logu.append(float(O.dm_log(np.array([((r1 >> 12) + 0.5) * 2.0 ** -52]))[0]))
    np.savez_compressed(OUT / "stats_and_draws.npz", chain=chain, rhat=O.rhat_gelman(chain), X=X, logp_mvn=O.logp(prob, X),
                        words=np.array(words, dtype=np.uint64), normals=np.array(normals), logu=np.array(logu))
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
