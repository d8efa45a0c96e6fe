"""ctypes face of oracle/demcz_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module (as the checker / the timed CPU baseline).  Nothing under ``demc.jl_amd/``
imports it.  Parity status against the Julia reference: "parity unpinned" (see the C header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"

SCHED_SYNCHRONOUS = 0
SCHED_SEQUENTIAL = 1
TARGET_MVNORMAL = 0
TARGET_ISO_QUAD = 1
TARGET_LINREG_SSE = 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class _Problem(C.Structure):
    _fields_ = [
        ("N", C.c_int64), ("chain_id0", C.c_int64), ("d", C.c_int32), ("K", C.c_int32),
        ("Mcap", C.c_int64), ("Nblocks", C.c_int32), ("block_offsets", _ip),
        ("block_indices", _ip), ("eps_scale", _dp), ("seed", C.c_uint64),
        ("target_kind", C.c_int32), ("mu", _dp), ("W", _dp), ("c0", C.c_double),
        ("design", _dp), ("yobs", _dp), ("nobs", C.c_int64),
    ]


def build(native: bool = False) -> Path:
    """Compile the oracle with gcc (a second or two).  ``native`` adds -O3 -march=native."""
    target = "native" if native else "all"
    subprocess.run(["make", "-C", str(_HERE), target], check=True, stdout=subprocess.DEVNULL)
    return _BUILD / ("libdemcz_oracle_native.so" if native else "libdemcz_oracle.so")


_LIBS: dict = {}


def lib(native: bool = False):
    key = bool(native)
    if key not in _LIBS:
        path = _BUILD / ("libdemcz_oracle_native.so" if native else "libdemcz_oracle.so")
        src = _HERE / "demcz_oracle.c"
        if not path.exists() or (src.exists() and src.stat().st_mtime > path.stat().st_mtime):
            build(native)
        L = C.CDLL(str(path))
        L.oracle_dm_log.restype = C.c_double
        L.oracle_dm_log.argtypes = [C.c_double]
        L.oracle_tempbaseline.restype = C.c_double
        L.oracle_tempbaseline.argtypes = [C.c_int64, C.c_int64, C.c_double, C.c_double]
        L.oracle_blocks_per_generation.restype = C.c_int64
        L.oracle_faithful_index_cost.restype = C.c_int64
        L.oracle_faithful_index_cost.argtypes = [C.c_int64, C.c_int64]
        _LIBS[key] = L
    return _LIBS[key]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t=_dp):
    return a.ctypes.data_as(t) if a is not None else None


class Problem:
    """Bundle of the sampler's static inputs; keeps the numpy buffers alive for ctypes."""

    def __init__(self, N, d, K, Mcap, eps_scale, seed, blocks=None, chain_id0=0, target=None):
        self.N, self.d, self.K, self.Mcap = int(N), int(d), int(K), int(Mcap)
        self.seed, self.chain_id0 = int(seed), int(chain_id0)
        if blocks is None:
            blocks = [list(range(d))]
        self.blocks = [list(map(int, b)) for b in blocks]
        offs = np.zeros(len(self.blocks) + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(b) for b in self.blocks])
        self._offs = offs
        self._idx = np.asarray([i for b in self.blocks for i in b], dtype=np.int32)
        self._eps = _f64(eps_scale)
        assert self._eps.shape == (d,)
        self.target = dict(target or {})
        kind = self.target.get("kind", "mvnormal")
        self._mu = self._W = self._design = self._y = None
        c0, nobs = 0.0, 0
        if kind == "mvnormal":
            tk = TARGET_MVNORMAL
            self._mu = _f64(self.target["mu"])
            self._W = np.asfortranarray(self.target["W"], dtype=np.float64)
            c0 = float(self.target["c0"])
        elif kind == "iso_quad":
            tk = TARGET_ISO_QUAD
            self._mu = _f64(self.target["mu"])
        elif kind == "linreg_sse":
            tk = TARGET_LINREG_SSE
            self._design = np.asfortranarray(self.target["design"], dtype=np.float64)
            self._y = _f64(self.target["y"])
            nobs = self._design.shape[0]
            assert self._design.shape[1] == d
        else:
            raise ValueError(kind)
        self.c = _Problem(self.N, self.chain_id0, self.d, self.K, self.Mcap, len(self.blocks),
                          _ptr(self._offs, _ip), _ptr(self._idx, _ip), _ptr(self._eps), self.seed, tk,
                          _ptr(self._mu), _ptr(self._W), c0, _ptr(self._design), _ptr(self._y), nobs)

    def blocks_per_generation(self, native=False):
        return int(lib(native).oracle_blocks_per_generation(C.byref(self.c)))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return list(o)


def draw_block(seed, chain, blk):
    o = (C.c_uint64 * 2)()
    lib().oracle_draw_block(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(blk), o)
    return int(o[0]), int(o[1])


def draw_rows(seed, chain, blk, M):
    """0-based archive rows (i1, i2) a block-step whose first Philox block is `blk` draws from M rows."""
    o = (C.c_uint64 * 2)()
    lib().oracle_draw_rows(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(blk), C.c_int64(M), o)
    return int(o[0]), int(o[1])


def dm_log(x):
    L = lib()
    x = np.asarray(x, dtype=np.float64)
    return np.array([L.oracle_dm_log(float(v)) for v in x.ravel()]).reshape(x.shape)


def dm_sincos2pi(k53):
    L = lib()
    c, s = C.c_double(), C.c_double()
    k53 = np.asarray(k53, dtype=np.uint64)
    out = np.empty(k53.shape + (2,))
    flat = out.reshape(-1, 2)
    for i, k in enumerate(k53.ravel()):
        L.oracle_dm_sincos2pi(C.c_uint64(int(k)), C.byref(c), C.byref(s))
        flat[i] = (c.value, s.value)
    return out


def normal_pair(r1, r2):
    z = (C.c_double * 2)()
    lib().oracle_normal_pair(C.c_uint64(r1), C.c_uint64(r2), z)
    return z[0], z[1]


def logp(prob: Problem, X):
    """X: (n, d) array -> logp (n,)."""
    X = np.asfortranarray(X, dtype=np.float64)
    n = X.shape[0]
    out = np.empty(n)
    lib().oracle_logp(C.byref(prob.c), _ptr(X), C.c_int64(n), C.c_int64(n), _ptr(out))
    return out


def block_step(prob: Problem, Z, M, chain_local, g, ib, gamma, x, lp, temperature=None):
    """One block-step with every intermediate.  Returns dict (fixture tier (i))."""
    Z = np.asfortranarray(Z, dtype=np.float64)
    assert Z.shape == (prob.Mcap, prob.d)
    x = _f64(x).copy()
    lpc = C.c_double(lp)
    b = len(prob.blocks[ib])
    nn = 1 if b == 1 else b
    dbg = np.zeros(5 + nn + prob.d)
    T = C.byref(C.c_double(temperature)) if temperature is not None else None
    acc = lib().oracle_block_step(C.byref(prob.c), _ptr(Z), C.c_int64(M), C.c_int64(chain_local),
                                  C.c_int64(g), C.c_int(ib), C.c_double(gamma), T, _ptr(x),
                                  C.byref(lpc), _ptr(dbg))
    return dict(i1=int(dbg[0]), i2=int(dbg[1]), logu=dbg[2], lp_prop=dbg[3], accepted=int(acc),
                normals=dbg[5:5 + nn].copy(), xprop=dbg[5 + nn:].copy(), x=x, logp=lpc.value)


def run(prob: Problem, X, lp, Z, M, g_from, g_to, gamma, temperature=None, schedule=SCHED_SYNCHRONOUS,
        do_append=True, history=True, native=False, rng_offset=0, threads=0):
    """Advance generations g_from..g_to (1-based, inclusive) IN PLACE on X (N,d) F-order,
    lp (N,), Z (Mcap,d) F-order.  Returns (M_new, chain (N,d,G) or None, log_obj (N,G) or None,
    changed (G,)).  threads > 0: the OpenMP loop over chains (synchronous schedule only; same bits)."""
    assert X.flags.f_contiguous and X.dtype == np.float64 and X.shape == (prob.N, prob.d)
    assert Z.flags.f_contiguous and Z.dtype == np.float64 and Z.shape == (prob.Mcap, prob.d)
    assert lp.dtype == np.float64 and lp.shape == (prob.N,)
    G = g_to - g_from + 1
    chain = np.zeros((prob.N, prob.d, G), order="F") if history else None
    lobj = np.zeros((prob.N, G), order="F") if history else None
    changed = np.zeros(G, dtype=np.int64)
    Mc = C.c_int64(M)
    temp = _f64(temperature) if temperature is not None else None
    if temp is not None:
        assert temp.shape == (G,)
    if threads > 0:
        assert schedule == SCHED_SYNCHRONOUS
        rc = lib(native).oracle_demcz_run_omp(C.byref(prob.c), _ptr(X), _ptr(lp), _ptr(Z), C.byref(Mc),
                                              C.c_int64(g_from), C.c_int64(g_to), C.c_double(gamma), _ptr(temp),
                                              _ptr(chain), _ptr(lobj), _ptr(changed, _lp),
                                              C.c_int(1 if do_append else 0), C.c_int64(rng_offset),
                                              C.c_int(threads))
    else:
        rc = lib(native).oracle_demcz_run(C.byref(prob.c), _ptr(X), _ptr(lp), _ptr(Z), C.byref(Mc),
                                          C.c_int64(g_from), C.c_int64(g_to), C.c_double(gamma), _ptr(temp),
                                          _ptr(chain), _ptr(lobj), _ptr(changed, _lp),
                                          C.c_int(schedule), C.c_int(1 if do_append else 0), C.c_int64(rng_offset))
    if rc != 0:
        raise RuntimeError(f"oracle_demcz_run failed rc={rc}")
    return int(Mc.value), chain, lobj, changed


def tempbaseline(ig, Ng, T0, TN):
    return lib().oracle_tempbaseline(int(ig), int(Ng), float(T0), float(TN))


def rhat_gelman(chain):
    chain = np.asfortranarray(chain, dtype=np.float64)
    N, d, G = chain.shape
    out = np.empty(d)
    rc = lib().oracle_rhat_gelman(_ptr(chain), C.c_int64(N), C.c_int64(G), C.c_int32(d), _ptr(out))
    if rc != 0:
        raise RuntimeError(f"oracle_rhat_gelman rc={rc}")
    return out


def changed_per_chain(log_obj):
    log_obj = np.asfortranarray(log_obj, dtype=np.float64)
    N, G = log_obj.shape
    out = np.zeros(N, dtype=np.int64)
    lib().oracle_changed_per_chain(_ptr(log_obj), C.c_int64(N), C.c_int64(G), _ptr(out, _lp))
    return out


def mean_cov_chain(chain):
    chain = np.asfortranarray(chain, dtype=np.float64)
    N, d, G = chain.shape
    mean = np.empty(d)
    cov = np.empty((d, d), order="F")
    lib().oracle_mean_cov_chain(_ptr(chain), C.c_int64(N), C.c_int64(G), C.c_int32(d), _ptr(mean), _ptr(cov))
    return mean, cov
