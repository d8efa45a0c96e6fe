"""Oracle log-densities against independent closed forms (scipy / numpy)."""
import numpy as np
import pytest
from scipy import stats


@pytest.mark.parametrize("d", [1, 2, 5, 10, 20])
def test_mvnormal_logpdf_vs_scipy(oracle, demc, d):
    """logpdf(MvNormal(mu, Sigma), x), test/example_normpdf.jl:13-16 (Distributions.jl is not
    vendored in the reference; scipy's closed form is the independent check)."""
    w = demc.workloads.mvnormal_problem(d, 8)
    prob = oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    rng = np.random.default_rng(d)
    X = w["mu"] + rng.standard_normal((200, d)) * 0.2
    got = oracle.logp(prob, X)
    ref = stats.multivariate_normal(w["mu"], w["Sigma"]).logpdf(X)
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-9)
    t = w["target"]
    assert np.allclose(t.W @ w["Sigma"] @ t.W.T, np.eye(d), atol=1e-10)
    assert np.allclose(np.triu(t.W, 1), 0)


def test_iso_quad_and_linreg(oracle, demc):
    w = demc.workloads.iso_quad_problem(10, 8)
    prob = oracle.Problem(8, 10, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    X = np.random.default_rng(0).standard_normal((50, 10))
    assert np.allclose(oracle.logp(prob, X), -((X - w["mu"]) ** 2).sum(axis=1), rtol=1e-13)
    w = demc.workloads.linreg_problem(10, 8, nobs=300)
    prob = oracle.Problem(8, 10, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    B = w["beta"] + 0.1 * np.random.default_rng(1).standard_normal((40, 10))
    ref = np.array([-0.5 * np.sum((w["y"] - w["design"] @ b) ** 2) for b in B])
    assert np.allclose(oracle.logp(prob, B), ref, rtol=1e-12)
