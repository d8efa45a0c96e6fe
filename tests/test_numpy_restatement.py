"""Oracle AND HIP path against the independent NumPy reading of the reference's Julia
(tests/golden/numpy_restatement.py -> tests/golden/np_*.npz, committed).

Tolerances (the restatement uses libm's log/sqrt/cos/sin and NumPy's sums and solve(Sigma, .); the oracle and
the kernels use the written-out arithmetic of DESIGN.md section 3):
    integers (Philox words, archive row indices, M, decisions, changed counts)   equal
    draws (normals, log u), proposals                                             rtol 1e-13
    log-densities                                                                 rtol 1e-11 (solve vs W factor)
    trajectories (chain, log_obj, Z) over <= 60 generations                       rtol 1e-10, atol 1e-13
    R-hat                                                                         rtol 1e-12
Every fixture was generated with no accept decision closer than 1e-9 to its threshold (stored as `margin`),
so rounding differences cannot flip a decision: a mismatch here is a structural one.
"""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

import demc_jl_amd as demc
from helpers import oracle_sample

GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))
TRAJ = sorted(p.stem for p in GOLD.glob("np_traj_*.npz"))
TOL = dict(rtol=1e-10, atol=1e-13)


def _target(g):
    if "target_Sigma" in g.files:
        return demc.MvNormalTarget(g["target_mu"], g["target_Sigma"])
    if "target_design" in g.files:
        return demc.LinRegSSETarget(g["target_design"], g["target_y"])
    return demc.IsoQuadTarget(g["target_mu"])


def _blocks(g):
    offs, idx = g["block_offsets"], g["block_indices0"]
    return [[int(v) for v in idx[offs[i]:offs[i + 1]]] for i in range(len(offs) - 1)]


def _close_traj(got_chain, got_lobj, got_Z, g):
    assert got_chain.shape == g["chain"].shape and got_Z.shape == g["Z"].shape, "shapes (M) differ"
    assert np.allclose(got_chain, g["chain"], **TOL)
    with np.errstate(invalid="ignore"):
        assert np.allclose(got_lobj, g["log_obj"], rtol=1e-10, atol=1e-11)
    assert np.allclose(got_Z, g["Z"], **TOL)


# ---- the restatement itself ---------------------------------------------------------------------------------
def test_restatement_philox_kat_and_fixture_is_reproducible():
    import numpy_restatement as R
    for ctr, key, out in R.KAT:
        assert R.philox4x32_10(ctr, key) == out
    g = np.load(GOLD / "np_traj_mvn_d5_N5_sync.npz")
    blocks1 = [[i + 1 for i in b] for b in _blocks(g)]
    r = R.demcz_generations(R.make_logobj("mvnormal", mu=g["target_mu"], Sigma=g["target_Sigma"]), g["Zinit"], int(g["N"]),
                            int(g["K"]), int(g["G"]), blocks1, g["eps_scale"], float(g["gamma"]), int(g["seed"]), "synchronous")
    assert np.array_equal(r["chain"], g["chain"]) and np.array_equal(r["Z"], g["Z"]) and float(g["margin"]) > 1e-9


# ---- oracle vs restatement (CPU) ----------------------------------------------------------------------------
def test_oracle_stream_words_rows_and_draws_equal_restatement(oracle):
    s = np.load(GOLD / "np_stats_and_draws.npz")
    for b in range(64):
        r1, r2 = oracle.draw_block(31953150, 5, 1000 + b)
        assert (r1, r2) == (int(s["words"][b, 0]), int(s["words"][b, 1]))
        z = oracle.normal_pair(r1, r2)
        assert np.allclose(z, s["normals"][b], rtol=1e-13, atol=0)
        assert np.allclose(oracle.dm_log(np.array([((r1 >> 12) + 0.5) * 2.0 ** -52]))[0], s["logu"][b], rtol=1e-13, atol=0)
    # archive rows for tiny, ordinary and > 2^31 archives: i1 != i2 always, both within 1..M
    Ms = (2, 3, 50, 1025, 2 ** 31 + 7)
    for k, (c, gidx) in enumerate(((0, 1), (3, 17), (1000, 99999))):
        for j, M in enumerate(Ms):
            i1, i2 = oracle.draw_rows(9, c, (gidx - 1) * 5, M)          # one block of 5 parameters: S = 1 + 3 + 1
            assert (i1 + 1, i2 + 1) == tuple(int(v) for v in s["rows_idx"][k, j])       # Julia's rows are 1-based
            assert i1 != i2 and 0 <= i1 < M and 0 <= i2 < M
    assert np.allclose(oracle.rhat_gelman(s["chain"]), s["rhat"], rtol=1e-12, atol=0)
    assert np.allclose(oracle.rhat_gelman(s["chain"][:, :, :40]), s["rhat_even"], rtol=1e-12, atol=0)
    assert np.allclose(s["rhat"], s["rhat_even"], rtol=0, atol=0), "utils.jl:4-8: an odd window drops its last sample"


def test_oracle_block_steps_equal_restatement(oracle):
    g = np.load(GOLD / "np_block_steps.npz")
    t = demc.MvNormalTarget(g["target_mu"], g["target_Sigma"])
    blocks = _blocks(g)
    prob = oracle.Problem(8, 5, 10, 50, g["eps_scale"], int(g["seed"]), blocks=blocks, target=t.spec())
    Z = np.asfortranarray(g["Z"])
    for i in range(int(g["n"])):
        T = float(g[f"T_{i}"])
        o = oracle.block_step(prob, Z, int(g["M"]), int(g[f"c_{i}"]), int(g[f"g_{i}"]), int(g[f"ib_{i}"]), float(g["gamma"]),
                              g[f"x0_{i}"], float(g[f"lp0_{i}"]), temperature=None if math.isnan(T) else T)
        assert (o["i1"] + 1, o["i2"] + 1) == (int(g[f"i1_{i}"]), int(g[f"i2_{i}"])), i      # Julia's rows are 1-based
        assert np.allclose(o["normals"], g[f"normals_{i}"], rtol=1e-13, atol=0), i
        assert np.allclose(o["logu"], g[f"logu_{i}"], rtol=1e-13, atol=0), i
        assert np.allclose(o["xprop"], g[f"xprop_{i}"], rtol=1e-13, atol=0), i
        assert np.allclose(o["lp_prop"], g[f"lp_prop_{i}"], rtol=1e-11, atol=0), i
        assert bool(o["accepted"]) == bool(g[f"accepted_{i}"]), i
        assert np.allclose(o["x"], g[f"x_{i}"], rtol=1e-13, atol=0) and np.allclose(o["logp"], g[f"logp_{i}"], rtol=1e-11, atol=0), i


@pytest.mark.parametrize("name", TRAJ)
def test_oracle_trajectories_equal_restatement(oracle, name):
    g = np.load(GOLD / f"{name}.npz")
    T = g["temperature"] if g["temperature"].size else None
    sched = 1 if str(g["schedule"]) == "sequential" else 0
    r = oracle_sample(oracle, _target(g), g["Zinit"], int(g["N"]), int(g["K"]), int(g["G"]), _blocks(g), g["eps_scale"],
                      float(g["gamma"]), int(g["seed"]), temperature=T, schedule=sched)
    assert r["M"] == g["Z"].shape[0]
    _close_traj(r["chain"], r["log_obj"], r["Z"], g)
    assert np.array_equal(r["changed"], g["changed"])


# ---- HIP path vs restatement (GPU; reads nothing but the .npz files) ----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 1])
@pytest.mark.parametrize("name", [n for n in TRAJ if not n.endswith("_seq")])
def test_hip_trajectories_equal_restatement(name, lanes):
    g = np.load(GOLD / f"{name}.npz")
    t, blocks = _target(g), _blocks(g)
    N, K, G = int(g["N"]), int(g["K"]), int(g["G"])
    kw = dict(verbose=False, seed=int(g["seed"]), lanes_per_chain=lanes)
    if g["temperature"].size == 0:
        mc, Z = demc.demcz_sample(t, g["Zinit"], N, K, G, len(blocks), blocks, g["eps_scale"], float(g["gamma"]), **kw)
    else:
        T = g["temperature"]
        mc, Z = demc.demcz_anneal(t, g["Zinit"], N, K, G, len(blocks), blocks, g["eps_scale"], float(g["gamma"]),
                                  temperaturefun=lambda ig, Ng, T0, TN: float(T[ig - 1]), adaptγ={"adapt": False}, **kw)
    _close_traj(mc.chain, mc.log_obj, Z, g)
    assert np.allclose(mc.Xcurrent, g["X"], **TOL)


@pytest.mark.gpu
def test_hip_draws_and_rhat_equal_restatement():
    s = np.load(GOLD / "np_stats_and_draws.npz")
    words, normals, logu = demc.selftest_draws(31953150, 5, 1000, 64)
    assert np.array_equal(words, s["words"])
    assert np.allclose(normals, s["normals"], rtol=1e-13, atol=0) and np.allclose(logu, s["logu"], rtol=1e-13, atol=0)
    assert np.allclose(demc.Rhat_gelman(s["chain"]), s["rhat"], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_hip_host_closure_block_steps_equal_restatement():
    """The host-closure path (demcz_propose / demcz_accept_commit, the split of update_demcz_chain_block at the
    closure call demcz.jl:189) against the restatement's block-steps: proposals, decisions and new states, with the
    restatement's own NumPy closure evaluating the log-density."""
    import numpy_restatement as R
    g = np.load(GOLD / "np_block_steps.npz")
    blocks = _blocks(g)
    logobj = R.make_logobj("mvnormal", mu=g["target_mu"], Sigma=g["target_Sigma"])
    N, d = 8, 5
    for i in range(int(g["n"])):
        c, gen, ib, T = int(g[f"c_{i}"]), int(g[f"g_{i}"]), int(g[f"ib_{i}"]), float(g[f"T_{i}"])
        e = demc.HipEngine(N=N, d=d, K=10 ** 9, Mcap=60, Gcap=0, blockindex=blocks, eps_scale=g["eps_scale"], seed=int(g["seed"]),
                           target=logobj)
        X = np.tile(g[f"x0_{i}"], (N, 1))
        e.set_state(X, np.full(N, float(g[f"lp0_{i}"])), g["Z"])
        Xp = e.propose(gen, ib, float(g["gamma"]))
        assert np.allclose(Xp[c], g[f"xprop_{i}"], rtol=1e-13, atol=0), i
        lp = np.array([logobj(Xp[k]) for k in range(N)])
        e.accept_commit(lp, None if math.isnan(T) else T)
        Xn, lpn, _, _ = e.get_state(with_Z=False)
        e.close()
        assert np.allclose(Xn[c], g[f"x_{i}"], rtol=1e-13, atol=0), i
        assert np.allclose(lpn[c], g[f"logp_{i}"], rtol=1e-11, atol=0), i
        assert (not np.array_equal(Xn[c], g[f"x0_{i}"])) == bool(g[f"accepted_{i}"]), i
