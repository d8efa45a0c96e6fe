"""Oracle Rhat_gelman / accept-ratio / mean_cov against a NumPy restatement of src/utils.jl."""
import numpy as np
import pytest


def rhat_numpy(chain):
    """Line-by-line NumPy restatement of Rhat_gelman, src/utils.jl:2-20."""
    Npop, Npar, Ngen = chain.shape
    n = Ngen // 2                                                   # :4
    m = Npop * 2                                                    # :5
    cs = np.zeros((m, Npar, n))
    cs[:Npop] = chain[:, :, :n]                                     # :7
    cs[Npop:] = chain[:, :, n:2 * n]                                # :8
    avg_par = cs.mean(axis=(0, 2), keepdims=True)                   # :10
    avg_chains = cs.mean(axis=2, keepdims=True)                     # :11
    B = n / (m - 1) * ((avg_chains - avg_par) ** 2).sum(axis=0)     # :13
    sj = 1 / (n - 1) * ((cs - avg_chains) ** 2).sum(axis=2, keepdims=True)   # :14
    W = 1 / m * sj.sum(axis=0)                                      # :15
    varhat = (n - 1) / n * W + 1 / n * B                            # :16
    return np.sqrt(varhat / W).ravel()                              # :18


@pytest.mark.parametrize("N,d,G", [(4, 5, 100), (5, 3, 101), (64, 2, 37), (3, 1, 4)])
def test_rhat_vs_numpy(oracle, N, d, G):
    rng = np.random.default_rng(N * 100 + G)
    chain = np.asfortranarray(np.cumsum(rng.standard_normal((N, d, G)), axis=2) * 0.1 + rng.standard_normal((N, d, 1)) + 5.0)
    assert np.allclose(oracle.rhat_gelman(chain), rhat_numpy(chain), rtol=1e-12)


def test_rhat_odd_window_drops_last_sample(oracle):
    rng = np.random.default_rng(3)
    chain = np.asfortranarray(rng.standard_normal((6, 2, 51)))
    a = oracle.rhat_gelman(chain)
    chain2 = chain.copy()
    chain2[:, :, 50] = 1e6                                          # utils.jl:4-8: never read
    assert np.array_equal(a, oracle.rhat_gelman(chain2))


def test_changed_per_chain_and_mean_cov(oracle):
    rng = np.random.default_rng(4)
    lo = rng.standard_normal((7, 60))
    keep = rng.random((7, 60)) < 0.7
    for g in range(1, 60):
        lo[keep[:, g], g] = lo[keep[:, g], g - 1]
    lo = np.asfortranarray(lo)
    ref = (np.diff(lo, axis=1) != 0).sum(axis=1)                    # utils.jl:61
    assert np.array_equal(oracle.changed_per_chain(lo), ref)
    chain = np.asfortranarray(rng.standard_normal((7, 3, 60)) + np.array([1.0, -2.0, 30.0])[None, :, None])
    mean, cov = oracle.mean_cov_chain(chain)
    flat = chain.transpose(1, 2, 0).reshape(3, -1)                  # flatten_chain, utils.jl:22-32
    b = flat.mean(axis=1)
    c = (flat - b[:, None]) @ (flat - b[:, None]).T / flat.shape[1]  # utils.jl:104
    assert np.allclose(mean, b, rtol=1e-13) and np.allclose(cov, c, rtol=1e-11, atol=1e-14)


def test_tempbaseline(oracle, demc):
    for ig, Ng, T0, TN in [(1, 1000, 3.0, 1e-3), (500, 1000, 1.0, 1e-3), (1000, 1000, 2.0, 1e-4), (7, 20000, 5.0, 0.0)]:
        ref = T0 * (TN / T0) ** (ig / Ng)                           # demcz_anneal.jl:1-3
        assert oracle.tempbaseline(ig, Ng, T0, TN) == pytest.approx(ref, rel=1e-15, abs=0)
        assert demc.tempbaseline(ig, Ng, T0, TN) == pytest.approx(ref, rel=1e-15, abs=0)
    assert demc.tempbaseline(1000, 1000, 3, 1e-3) == pytest.approx(1e-3)
