"""GPU: two chains to a wave (window_kernel_ps2d, demcz_kernels_ps2d.h).  Lanes 32..63 of the steady-state wave-per-chain consumer
used to shadow lanes 0..31; here they run a second chain -- the LIVE launch, whose waves must all be resident, then holds 2048
chains where it held 1024, at the same time per launch.  Same arithmetic on the same values as the one-chain kernel: every case
is compared with the oracle bit for bit (update_demcz_chain_block / accept / runchain!, src/demcz.jl:80-93, 174-203)."""
import os

import numpy as np
import pytest

from helpers import SPLIT_WAVE, oracle_sample

pytestmark = pytest.mark.gpu

THREADS = max(1, min(len(os.sched_getaffinity(0)), 8))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def _run(demc, w, N, d, K, G, seed, pieces=None, temperature=None, spin=0, fault=None):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    if spin:
        e.set_live_spin_limit(spin)
    if fault is not None:
        e.debug_set_live_fault(-1, fault)
    g = 1
    names = set()
    for n in (pieces or [G]):
        e.run(g, g + n - 1, w["gamma"], None if temperature is None else temperature[g - 1:g + n - 1])
        names.add(e.kernel_name().split("<")[0])
        g += n
    e.synchronize()
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    out = dict(chain=ch, log_obj=lo, X=X, logp=lp, Z=Z, M=M, lanes=e.info()["lanes_per_chain"], launches=e.info()["window_launches"],
               live=e.live_status(), kernels=names, counts=e.kernel_counts(), changed_total=e.changed_total(1, G))
    e.close()
    return out


def _same(a, ref):
    assert np.array_equal(a["chain"], ref["chain"]), "chain history differs from the oracle"
    assert np.array_equal(a["log_obj"], ref["log_obj"])
    assert np.array_equal(a["X"], ref["X"]) and np.array_equal(a["logp"], ref["logp"])
    assert a["M"] == ref["M"] and np.array_equal(a["Z"], ref["Z"]), "archive differs from the oracle"
    assert a["changed_total"] == int(np.sum(ref["changed"]))


@pytest.mark.parametrize("N", [2048, 1536])
def test_two_chains_per_wave_is_chosen_beyond_1024_chains_and_equals_oracle(demc, oracle, N):
    """What the library chooses by itself at d = 5, K = 10 once one chain per wave no longer fits a LIVE launch: the two-chain
    kernel, LIVE, a handful of launches for 1000 generations."""
    d, K, G, seed = 5, 10, 1000, 700 + N
    w = demc.workloads.mvnormal_problem(d, N)
    a = _run(demc, w, N, d, K, G, seed)
    assert a["lanes"] == SPLIT_WAVE and a["live"] == (True, 0) and a["launches"] <= 4, (a["lanes"], a["live"], a["launches"])
    assert a["kernels"] == {"demcz::window_kernel_ps2d"}, a["kernels"]
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _same(a, ref)


@pytest.mark.parametrize("N,d,K,pieces", [(37, 5, 10, [300]), (1, 3, 5, [100]), (2, 2, 5, [60]), (255, 4, 15, [150]),
                                          (64, 5, 10, [5, 7, 203, 85]), (33, 5, 5, [100, 3, 97])])
def test_two_chains_per_wave_odd_populations_and_irregular_pieces(demc, oracle, monkeypatch, N, d, K, pieces):
    """Forced at any N (DEMCZ_PS_DUAL): odd populations (the last wave's second half shadows the last chain and writes nothing),
    every d of the layout, and calls whose pieces are not whole passes -- those launches take the general one-chain kernel,
    one per K-window, between the two-chain ones."""
    monkeypatch.setenv("DEMCZ_PS_DUAL", "1")
    G, seed = sum(pieces), 90 + N + d
    w = demc.workloads.mvnormal_problem(d, N)
    a = _run(demc, w, N, d, K, G, seed, pieces=pieces)
    assert a["lanes"] == SPLIT_WAVE and a["live"][1] == 0
    assert "demcz::window_kernel_ps2d" in a["kernels"]
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    _same(a, ref)


def test_two_chains_per_wave_tempered_and_redone(demc, oracle, monkeypatch):
    """The tempered accept (demcz_anneal.jl:172-178) in the two-chain kernel; a forced hand-off time-out (poll limit 1) and a launch
    that finds the error word set (the snapshot rows of BOTH chains of every wave must have been written) are redone bit-exactly."""
    monkeypatch.setenv("DEMCZ_PS_DUAL", "1")
    N, d, K, G, seed = 301, 5, 5, 200, 17
    w = demc.workloads.mvnormal_problem(d, N)
    T = np.array([demc.tempbaseline(g, G, 3.0, 1e-3) for g in range(1, G + 1)])
    T[[20, 21, 150]] = 0.0
    a = _run(demc, w, N, d, K, G, seed, temperature=T)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, temperature=T)
    _same(a, ref)
    refp = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    b = _run(demc, w, N, d, K, G, seed, spin=1)
    assert b["live"] == (False, 1)
    _same(b, refp)
    c = _run(demc, w, N, d, K, G, seed, fault=1)
    assert c["live"] == (False, 1) and np.isfinite(c["X"]).all()
    _same(c, refp)
