"""Synthetic inputs of the BASELINE configs C1-C5.

Everything is generated from ``numpy.random.Generator(numpy.random.Philox(seed))`` with the
reference's own test seed 31953150 (test/example_normpdf.jl:6); Julia's MersenneTwister stream
itself cannot be regenerated without Julia, so these are inputs of the same *shape and
distribution*, not the same numbers.
"""
from __future__ import annotations

import numpy as np

from .targets import MvNormalTarget, IsoQuadTarget, LinRegSSETarget

SEED = 31953150
SEED_LINREG = 319531501     # test/example_linreg.jl:6


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def mvnormal_problem(d, N, seed=SEED):
    """test/example_normpdf.jl:8-13, 20-26: mu ~ U(0,1)^d, A ~ U(0,1)^(d x d),
    Sigma = A'A + 2I, Sigma ./ maximum(Sigma) / 100; eps_scale = 1e-5, gamma = 2.38, K = 10,
    Z ~ N(0,1)^(M0 x d) with M0 = max(10 d, N)."""
    r = _rng(seed)
    mu = r.random(d)
    A = r.random((d, d))
    Sigma = A.T @ A + 2.0 * np.eye(d)
    Sigma = Sigma / Sigma.max() / 100.0
    M0 = max(10 * d, N)
    Z = np.asfortranarray(r.standard_normal((M0, d)))
    return dict(target=MvNormalTarget(mu, Sigma), mu=mu, Sigma=Sigma, Zinit=Z, eps_scale=1e-5 * np.ones(d),
                gamma=2.38, K=10, d=d, N=N)


def iso_quad_problem(d, N, seed=SEED):
    """test/test_anneal.jl:7-20."""
    r = _rng(seed)
    mu = r.random(d)
    M0 = max(10 * d, N)
    Z = np.asfortranarray(r.standard_normal((M0, d)))
    return dict(target=IsoQuadTarget(mu), mu=mu, Zinit=Z, eps_scale=1e-5 * np.ones(d), gamma=2.38, K=10, d=d, N=N)


def linreg_problem(d, N, nobs=1000, seed=SEED_LINREG):
    """test/example_linreg.jl:9-32 restated at d parameters (intercept + d-1 regressors with
    unit variance and 0.25 covariance), beta = 1 + 3 U(0,1)^d, y = X beta + N(0,1)."""
    r = _rng(seed)
    npar = d - 1
    Sx = np.full((npar, npar), 0.25) + 0.75 * np.eye(npar)
    Lx = np.linalg.cholesky(Sx)
    X = np.ones((nobs, d))
    X[:, 1:] = r.standard_normal((nobs, npar)) @ Lx.T
    beta = 1.0 + 3.0 * r.random(d)
    y = X @ beta + r.standard_normal(nobs)
    M0 = max(10 * d, N)
    Z = np.asfortranarray(r.standard_normal((M0, d)))
    return dict(target=LinRegSSETarget(X, y), design=X, y=y, beta=beta, Zinit=Z, eps_scale=1e-5 * np.ones(d),
                gamma=2.0, K=10, d=d, N=N)


CONFIGS = {
    "C1": dict(kind="mvnormal", d=5, N=4, Ngeneration=10000, blocks=None),
    "C2": dict(kind="mvnormal", d=5, N=1024, Ngeneration=10000, blocks=None),
    "C3": dict(kind="mvnormal", d=20, N=4096, Ngeneration=10000, blocks=[range(0, 5), range(5, 10), range(10, 15), range(15, 20)]),
    "C4": dict(kind="mvnormal", d=20, N=8192, Ngeneration=10000, blocks=None),
    "C5": dict(kind="linreg", d=10, N=2048, Ngeneration=10000, blocks=None),
}
