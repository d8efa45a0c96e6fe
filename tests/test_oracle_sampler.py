"""The reference's own statistical predicates (test/example_normpdf.jl:49-51), run on the oracle:
all(Rhat .< 1.1), all(0.1 .< accept_ratio .< 0.45) over the last 2500 of 10000 generations with
N=5 chains, d=5 -- for both update schedules (reference Gauss-Seidel order and the GPU's
synchronous order) -- plus the moment check the reference only prints (:42)."""
import numpy as np
import pytest

import demc_jl_amd as demc
from helpers import oracle_sample


@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("N", [5, 4])
def test_reference_predicates_example_normpdf(oracle, schedule, N):
    d, G = 5, 10000
    w = demc.workloads.mvnormal_problem(d, N)
    Z0 = w["Zinit"][:10 * d]                                         # Z = randn(10*ndim, ndim), :26
    r = oracle_sample(oracle, w["target"], Z0, N, 10, G, None, w["eps_scale"], 2.38, seed=31953150 + schedule,
                      schedule=schedule)
    keep = slice(G - 2500, G)                                        # :35-39
    chain, lobj = r["chain"][:, :, keep], r["log_obj"][:, keep]
    Rhat = oracle.rhat_gelman(chain)
    acc = oracle.changed_per_chain(lobj) / (lobj.shape[1] - 1)
    assert np.all(Rhat < 1.1), Rhat                                  # :49
    assert np.all(acc > 0.1) and np.all(acc < 0.45), acc             # :50-51
    mean, cov = oracle.mean_cov_chain(chain)
    sd = np.sqrt(np.diag(w["Sigma"]))
    assert np.all(np.abs(mean - w["mu"]) < 0.25 * sd)                # few effective samples at N=5
    assert r["M"] == Z0.shape[0] + N * (G // 10)


def test_schedules_agree_in_distribution(oracle):
    """Sequential (reference) and synchronous (GPU) schedules sample the same target."""
    d, N, G = 5, 64, 3000
    w = demc.workloads.mvnormal_problem(d, N)
    out = []
    for schedule in (0, 1):
        r = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, seed=11, schedule=schedule)
        mean, cov = oracle.mean_cov_chain(r["chain"][:, :, 1000:])
        out.append((mean, cov))
        sd = np.sqrt(np.diag(w["Sigma"]))
        assert np.all(np.abs(mean - w["mu"]) < 0.1 * sd)
        assert np.allclose(cov, w["Sigma"], rtol=0.15, atol=0.1 * w["Sigma"].max())
    assert not np.array_equal(out[0][0], out[1][0])                  # they are different trajectories


def test_window_split_invariance(oracle):
    """Running generations in one call or in arbitrary pieces gives identical bits: draws are
    keyed by (chain, generation), not by call order."""
    d, N, G = 5, 16, 47
    w = demc.workloads.mvnormal_problem(d, N)
    a = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, seed=5)
    prob = a["prob"]
    M0 = w["Zinit"].shape[0]
    X = np.array(w["Zinit"][M0 - N:], order="F")
    lp = oracle.logp(prob, X)
    Z = np.zeros((prob.Mcap, d), order="F")
    Z[:M0] = w["Zinit"]
    M, g, chains = M0, 1, []
    for step in (3, 7, 10, 1, 19, 7):
        M, ch, lo, _ = oracle.run(prob, X, lp, Z, M, g, g + step - 1, 2.38)
        chains.append(ch)
        g += step
    assert np.array_equal(np.concatenate(chains, axis=2), a["chain"]) and M == a["M"]


def test_openmp_loop_is_bit_identical(oracle):
    """The CPU-omp baseline row (oracle_demcz_run_omp) is the synchronous schedule spread over threads:
    same streams, same arithmetic, same bits for any thread count."""
    from demc_jl_amd import workloads
    N, d, G, K = 24, 5, 40, 10
    w = workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    outs = []
    for thr in (0, 1, 3):
        Mcap = M0 + N * G // K
        prob = oracle.Problem(N, d, K, Mcap, w["eps_scale"], 7, target=w["target"].spec())
        X = np.array(w["Zinit"][-N:], order="F"); lp = oracle.logp(prob, X)
        Z = np.zeros((Mcap, d), order="F"); Z[:M0] = w["Zinit"]
        M, chain, lobj, changed = oracle.run(prob, X, lp, Z, M0, 1, G, 2.38, threads=thr)
        outs.append((M, chain, lobj, changed, Z.copy()))
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        for a, b in zip(o[1:], outs[0][1:]):
            assert np.array_equal(a, b)
