"""GPU: the regression target's two-generations-per-pass kernel (window_kernel_lr8s, demcz_kernels_lr.h).

Eight chains per workgroup; a chain's next generation and the one after it (as if the first were rejected) share one
log-density pass, so chains of a workgroup advance at their own pace, by one or two generations a step.  Everything that
makes that different from the lock-step kernels is exercised here against the oracle, bit for bit: acceptance ratios from a
few per cent to most proposals (how often the second column is thrown away), K from 1 (no second column ever crosses a
boundary) upward, populations that do not fill the last workgroup, tempering down to T = 0, the deferred schedule
(one launch per batch, no in-launch hand-off), and the forced hand-off timeout."""
import numpy as np
import pytest

import demc_jl_amd as demc_pkg
from helpers import SPLIT, oracle_sample
from oracle_engine import OracleEngine

pytestmark = pytest.mark.gpu


def _engine(demc, w, N, d, K, G, seed, lanes=SPLIT):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)],
                       eps_scale=w["eps_scale"], seed=seed, target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    return e


def _same(e, ref, G):
    chain, lobj = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    assert np.array_equal(chain, ref["chain"]) and np.array_equal(lobj, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("N,K,G,gamma,temper", [
    (64, 10, 200, 2.38, None),      # few acceptances: almost every step resolves two generations
    (64, 10, 200, 0.3, None),       # many acceptances: the second column is often discarded, chains drift apart
    (61, 7, 150, 0.6, None),        # the last workgroup is not full; K not a multiple of anything convenient
    (24, 1, 90, 0.6, None),         # every generation is a boundary: never a second column
    (40, 2, 120, 0.3, None),        # a boundary every other generation
    (128, 10, 300, 0.6, "anneal"),  # tempered, T from 3 to 1e-3
    (32, 5, 100, 0.6, "zero"),      # T = 0 from the first generation (demcz_anneal.jl:18 default): Inf / NaN comparisons
    (2048, 10, 60, 0.6, None),      # C5's population: all 256 workgroups
])
def test_two_generations_per_pass_bit_exact(demc, oracle, N, K, G, gamma, temper):
    d, seed = 10, 77
    w = demc.workloads.linreg_problem(d, N, nobs=90)
    T = None
    if temper == "anneal":
        T = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])
    elif temper == "zero":
        T = np.zeros(G)
    e = _engine(demc, w, N, d, K, G, seed)
    cut = G // 3
    e.run(1, cut, gamma, None if T is None else T[:cut])       # (two calls: the second starts mid-window)
    e.run(cut + 1, G, gamma, None if T is None else T[cut:])
    e.synchronize()
    on, redos = e.live_status()
    assert redos == 0
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], gamma, seed, temperature=T)
    _same(e, ref, G)
    for a, b in [(1, G), (cut + 1, G), (cut + 2, G)]:           # the ballot counts of both columns (whole launches, -1st generation)
        tot, from_ballots = e.changed_total(a, b, with_source=True)
        assert from_ballots and tot == int(ref["changed"][a - 1:b].sum()), (a, b)
    e.close()


@pytest.mark.parametrize("K", [1, 3])
def test_forced_handoff_timeout_is_redone_bit_exact(demc, oracle, K):
    """Poll limit 1: a chain that finds a row missing once gives the launch up; everything is redone one K-window at a time
    (the same kernel, not LIVE) and equals the oracle."""
    N, d, G, seed, gamma = 512, 10, 120, 41, 0.6
    w = demc.workloads.linreg_problem(d, N, nobs=70)
    e = _engine(demc, w, N, d, K, G, seed)
    e.set_live_spin_limit(1)
    e.set_live_rearms(0)                  # (the fall-back for good; re-arming: tests/test_gpu_live.py)
    e.run(1, 50, gamma)
    e.run(51, G, gamma)
    e.synchronize()
    on, redos = e.live_status()
    assert redos == 1 and not on, "the poll limit of 1 must have forced the fall-back"
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], gamma, seed)
    _same(e, ref, G)
    e.set_live_spin_limit(0)
    e.close()


@pytest.mark.parametrize("E", [1, 3])
def test_deferred_schedule_runs_the_same_kernel_without_handoff(E):
    """append_lag > 0: one launch per batch of E boundaries, rows visible E windows later -- the non-LIVE instantiation,
    single handle and two in-process shards, against the oracle-backed emulation of the same schedule."""
    N, d, G = 64, 10, 95
    w = demc_pkg.workloads.linreg_problem(d, N, nobs=50)

    def run(engine_factory, sharding):
        opts = demc_pkg.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False)
        opts.γ = 0.6
        return demc_pkg.demcz_sample(w["target"], w["Zinit"], opts, seed=9, engine_factory=engine_factory, sharding=sharding,
                                     append_lag=E, lanes_per_chain=SPLIT)
    ref, Zref = run(OracleEngine, demc_pkg.Sharding(mode="host", local_shards=1, host_exchange_always=True))
    a, Za = run(None, None)
    assert np.array_equal(a.chain, ref.chain) and np.array_equal(a.log_obj, ref.log_obj) and np.array_equal(Za, Zref)
    b, Zb = run(None, demc_pkg.Sharding(mode="host", local_shards=2))
    assert np.array_equal(b.chain, ref.chain) and np.array_equal(Zb, Zref)
