"""Device-evaluated log-densities.

The reference takes an arbitrary Julia closure ``logobj(x)::Float64`` (src/demcz.jl:189).  The
three closures its tests and examples build are available as device targets, evaluated inside
the chain-update kernel; any other Python callable is driven through the host-closure mode
(``demcz_propose`` / ``demcz_accept_commit``), one batch of N proposals per round trip.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class MvNormalTarget:
    """``logpdf(MvNormal(mu, Sigma), x)`` -- test/example_normpdf.jl:13-16, README.md:27-28.

    Evaluated as ``c0 - 0.5 * ||W (x - mu)||^2`` with ``W = inv(chol(Sigma))`` (lower
    triangular) and ``c0 = -0.5 (d log 2pi + logdet Sigma)``; ``W`` and ``c0`` are computed
    here once, in float64, and are inputs to the device.
    """
    mu: np.ndarray
    Sigma: np.ndarray

    def __post_init__(self):
        self.mu = np.ascontiguousarray(self.mu, dtype=np.float64)
        self.Sigma = np.ascontiguousarray(self.Sigma, dtype=np.float64)
        d = self.mu.shape[0]
        if self.Sigma.shape != (d, d):
            raise ValueError("Sigma must be d x d")
        L = np.linalg.cholesky(self.Sigma)
        # forward substitution L W = I, row by row (W is lower triangular)
        W = np.zeros((d, d))
        for j in range(d):
            for i in range(j, d):
                s = (1.0 if i == j else 0.0) - float(np.dot(L[i, j:i], W[j:i, j]))
                W[i, j] = s / L[i, i]
        self.W = np.asfortranarray(W)
        self.c0 = -0.5 * (d * math.log(2.0 * math.pi) + 2.0 * float(np.sum(np.log(np.diag(L)))))
        self.d = d

    kind = _lib.TARGET_MVNORMAL

    def fill(self, cfg, keep):
        keep += [self.mu, self.W]
        cfg.mu, cfg.W, cfg.c0 = _lib.ptr(self.mu), _lib.ptr(self.W), self.c0

    def spec(self):
        return dict(kind="mvnormal", mu=self.mu, W=self.W, c0=self.c0)


@dataclass
class IsoQuadTarget:
    """``-sum((x .- mu).^2)`` -- test/test_anneal.jl:10."""
    mu: np.ndarray

    def __post_init__(self):
        self.mu = np.ascontiguousarray(self.mu, dtype=np.float64)
        self.d = self.mu.shape[0]

    kind = _lib.TARGET_ISO_QUAD

    def fill(self, cfg, keep):
        keep += [self.mu]
        cfg.mu = _lib.ptr(self.mu)

    def spec(self):
        return dict(kind="iso_quad", mu=self.mu)


@dataclass
class LinRegSSETarget:
    """``-0.5 * sum((y .- X*b).^2)`` -- test/example_linreg.jl:32.  ``design`` is nobs x d."""
    design: np.ndarray
    y: np.ndarray

    def __post_init__(self):
        self.design = np.asfortranarray(self.design, dtype=np.float64)
        self.y = np.ascontiguousarray(self.y, dtype=np.float64)
        if self.design.shape[0] != self.y.shape[0]:
            raise ValueError("design rows != len(y)")
        self.d = self.design.shape[1]

    kind = _lib.TARGET_LINREG_SSE

    def fill(self, cfg, keep):
        keep += [self.design, self.y]
        cfg.design, cfg.yobs, cfg.nobs = _lib.ptr(self.design), _lib.ptr(self.y), self.design.shape[0]

    def spec(self):
        return dict(kind="linreg_sse", design=self.design, y=self.y)


def is_device_target(obj) -> bool:
    return isinstance(obj, (MvNormalTarget, IsoQuadTarget, LinRegSSETarget))
