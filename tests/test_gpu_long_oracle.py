"""Long runs at the BASELINE shapes compared with the CPU ORACLE directly (not with another HIP kernel), bit for bit:
the wave-per-chain consumer at d = 20 / 10 / 8 (demcz_kernels_pw.h: LDS-DMA with hand-counted waits, LIVE hand-off),
the regression kernel with two generations per pass at C5's real shape (demcz_kernels_lr.h, window_kernel_lr8s), and the
four-wave block-update kernel at C3 (demcz_kernels_ml.h, window_kernel_mlb).  What they stand for in the reference:
src/demcz.jl:80-93, 167-195 and src/demcz_anneal.jl:149-178.  The oracle runs its OpenMP loop over chains (same bits as its
single-thread loop: tests/test_oracle_sampler.py), so each case costs a few seconds of host time."""
import functools
import os

import numpy as np
import pytest

from helpers import SPLIT, SPLIT_WAVE, oracle_sample

pytestmark = pytest.mark.gpu

THREADS = max(1, min(len(os.sched_getaffinity(0)), 8))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
BLOCKS_D20 = [list(range(0, 5)), list(range(5, 10)), list(range(10, 15)), list(range(15, 20))]


def _hip(demc, w, N, d, K, G, blocks, seed, gamma, lanes=0, temperature=None, pieces=None, lag=0):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], lanes_per_chain=lanes)
    if lag:
        e.set_append_lag(lag)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    g = 1
    for n in (pieces or [G]):
        e.run(g, g + n - 1, gamma, None if temperature is None else temperature[g - 1:g + n - 1])
        g += n
    assert g == G + 1
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    info, live = e.info(), e.live_status()
    tot = e.changed_total(1, G)
    e.close()
    return dict(chain=ch, log_obj=lo, X=X, logp=lp, Z=Z, M=M, lanes=info["lanes_per_chain"], launches=info["window_launches"],
                live=live, changed_total=tot)


def _same(a, ref):
    assert np.array_equal(a["chain"], ref["chain"]), "chain history differs from the oracle"
    assert np.array_equal(a["log_obj"], ref["log_obj"]), "log_obj history differs from the oracle"
    assert np.array_equal(a["X"], ref["X"]) and np.array_equal(a["logp"], ref["logp"])
    assert a["M"] == ref["M"] and np.array_equal(a["Z"], ref["Z"]), "archive differs from the oracle"
    assert a["changed_total"] == int(np.sum(ref["changed"]))


@pytest.mark.parametrize("d,G", [(20, 3000), (10, 2000), (8, 2000)])
def test_wave_per_chain_live_long_run_equals_oracle(demc, oracle, d, G):
    """window_kernel_pw<0, d, LIVE> at N = 1024, K = 10 (d = 20: C4's per-GPU shard): a few launches of up to ~370-1000
    generations each, rows handed from wave to wave inside them."""
    N, K, seed = 1024, 10, 4100 + d
    w = demc.workloads.mvnormal_problem(d, N)
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, w["gamma"], pieces=[G // 3, 1, G - G // 3 - 1])
    assert a["lanes"] == SPLIT_WAVE and a["live"] == (True, 0), (a["lanes"], a["live"])
    assert a["launches"] < G // K // 4              # LIVE launches: far fewer than one per K-window
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _same(a, ref)


@pytest.mark.parametrize("kind,d", [("mvn", 6), ("mvn", 7), ("mvn", 12), ("mvn", 16), ("mvn", 22), ("mvn", 23), ("mvn", 26), ("mvn", 30), ("mvn", 32),
                                    ("iso", 10), ("iso", 30)])
def test_wave_per_chain_every_dimension_live_equals_oracle(demc, oracle, kind, d):
    """Round 5: window_kernel_pw is no longer built for d = 8 / 10 / 20 only.  The reference's own scripts run at d = 10 (iso-quad,
    test/test_anneal.jl:7-10), 26 (test/example_linreg.jl:9) and 30 (test/test_anneal_parallel.jl:16); README.md:16 says "10-20
    dimensions".  N = 1024 chains, K = 10, the library's choice of layout: one wave per chain, LIVE launches (from d = 21 on one
    workgroup per CU), regular and irregular pieces; the isotropic quadratic tempered (the annealer's use of it)."""
    N, K, seed = 1024, 10, 5200 + d
    G = 600 if d <= 16 else 400
    w = demc.workloads.mvnormal_problem(d, N) if kind == "mvn" else demc.workloads.iso_quad_problem(d, N)
    T = None if kind == "mvn" else np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])
    gamma = w["gamma"]
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, gamma, temperature=T, pieces=[G // 2, 3, G - G // 2 - 3])
    assert a["lanes"] == SPLIT_WAVE and a["live"] == (True, 0), (a["lanes"], a["live"])
    assert a["launches"] <= 12, a["launches"]
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], gamma, seed, temperature=T, threads=THREADS)
    _same(a, ref)


@pytest.mark.parametrize("d,N,K,temper", [(20, 1024, 10, False), (20, 200, 5, True), (10, 1024, 15, False), (8, 333, 10, True)])
def test_wave_per_chain_regular_launches_equal_oracle(demc, oracle, d, N, K, temper):
    """Round 4: LIVE launches that start behind a K boundary, with K and their length multiples of five, take the regular form of
    window_kernel_pw (REG: every pass five generations, a boundary counter instead of the queue of pass lengths); the others keep
    the general form.  Both against the oracle, in one run: regular pieces, then a piece that is not a multiple of five, then one
    that starts inside a K-window."""
    G, seed = 40 * K + 7, 9100 + d
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    T = None
    if temper:
        T = np.array([demc.tempbaseline(g, 100, 3, 1e-2) for g in range(1, G + 1)])
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], lanes_per_chain=SPLIT_WAVE)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    names = []
    g = 1
    for n in (20 * K, 10 * K, 7, 10 * K):
        e.run(g, g + n - 1, w["gamma"], None if T is None else T[g - 1:g + n - 1])
        names.append(e.kernel_name())
        g += n
    assert g == G + 1
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    live = e.live_status()
    e.close()
    assert live == (True, 0), live
    reg = [n.count(",") == 5 and n.endswith("false, true>") for n in names]       # <TARGET, D, LIVE, TEMPER, MF = false, REG = true>
    assert reg == [True, True, False, False], names
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, temperature=T, threads=THREADS)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


def test_c4_shard_deferred_visibility_long_run_equals_oracle(demc, oracle):
    """C4's per-GPU shard with append_lag = 2 (the non-LIVE instantiation of window_kernel_pw, one launch per two K-windows, its
    producer half riding in the same grid) against the oracle-backed emulation of the same visibility rule."""
    from oracle_engine import OracleEngine
    d, N, K, G, E, seed = 20, 1024, 10, 1000, 2, 77
    w = demc.workloads.mvnormal_problem(d, N)
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, w["gamma"], lag=E, pieces=[333, 667])
    assert a["lanes"] == SPLIT_WAVE
    opts = demc.demcopt(d, N=N, K=K, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="no")
    sh = demc.Sharding(mode="host", local_shards=1, host_exchange_always=True)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=seed, sharding=sh, append_lag=E,
                              engine_factory=functools.partial(OracleEngine, threads=THREADS))
    assert np.array_equal(a["chain"], mc.chain) and np.array_equal(a["log_obj"], mc.log_obj)
    assert np.array_equal(a["X"], mc.Xcurrent) and np.array_equal(a["Z"], Z)


@pytest.mark.parametrize("gamma,G", [(0.5, 1000), (2.0, 400)])
def test_c5_regression_two_generations_per_pass_long_run_equals_oracle(demc, oracle, gamma, G):
    """window_kernel_lr8s<10, LIVE> at C5's real shape (nobs = 1000, N = 2048, tempered).  gamma = 0.5 gives the ~22 %
    acceptance the annealer's adaptation steers for (second columns are discarded often); gamma = 2.0 is
    test/example_linreg.jl's (~1 %: nearly every pass resolves two generations)."""
    d, N, K, seed = 10, 2048, 10, 319531501
    w = demc.workloads.linreg_problem(d, N)
    T = np.array([demc.tempbaseline(g, 10000, 3, 1e-3) for g in range(1, G + 1)])
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, gamma, temperature=T, pieces=[G // 2 + 3, G - G // 2 - 3])
    assert a["lanes"] == SPLIT
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], gamma, seed, temperature=T, threads=THREADS)
    _same(a, ref)
    acc = np.mean(ref["changed"][1:]) / N
    assert (0.1 < acc < 0.45) if gamma == 0.5 else (acc < 0.1), acc


@pytest.mark.parametrize("gamma,G", [(0.5, 300), (2.38, 200)])
def test_regression_at_the_reference_example_shape_equals_oracle(demc, oracle, gamma, G):
    """Round 5: test/example_linreg.jl's own shape -- 25 regressors + intercept = 26 parameters, nobs = 1000 (:9-32) -- at N = 1024
    chains, annealed: window_kernel_ml<LINREG_SSE, 26, 16, ..., COOP> (sixteen lanes per chain, seven helper waves per chain wave
    forming the residuals from tiles of the design, proposals by DPP, the chain wave's fold in the spec's order), cut into
    irregular pieces, at an acceptance the annealer steers for and at the example's gamma."""
    d, N, K, seed = 26, 1024, 10, 2605
    w = demc.workloads.linreg_problem(d, N, nobs=1000)
    T = np.array([demc.tempbaseline(g, 5000, 3, 1e-3) for g in range(1, G + 1)])
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, gamma, temperature=T, pieces=[G // 2 + 7, 1, G - G // 2 - 8])
    assert a["lanes"] == 16, a["lanes"]
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], gamma, seed, temperature=T, threads=THREADS)
    _same(a, ref)


def test_c3_block_updates_long_run_equals_oracle(demc, oracle):
    """window_kernel_mlb<0, 20, 16, REC, LIVE> with four-wave workgroups at C3 (d = 20 in four blocks of five, N = 4096)."""
    d, N, K, G, seed = 20, 4096, 10, 1000, 31953150
    w = demc.workloads.mvnormal_problem(d, N)
    a = _hip(demc, w, N, d, K, G, BLOCKS_D20, seed, w["gamma"], pieces=[211, 789])
    assert a["lanes"] == SPLIT
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, BLOCKS_D20, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _same(a, ref)


def test_c2_eighty_autostop_slabs_equal_oracle(demc, oracle):
    """The bench's workload for 80 slabs -- 8.2e7 chain-updates through demcz_run_checked (window_kernel_ps2 LIVE launches of
    1000 generations, the producer's lane-per-generation records, the R-hat monitor beside them), the archive growing to
    8.2 M rows -- against the oracle: history, state, archive and the R-hat of every slab.  demcz.jl:30-55."""
    N, d, K, every, S, seed = 1024, 5, 10, 1000, 80, 31953150
    G = S * every
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    g_stop, rmax, _ = e.run_checked(1, G, w["gamma"], every, 0.0)
    assert g_stop == G and len(rmax) == S
    counts, live = e.kernel_counts(), e.live_status()
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    assert counts["ps2"] == S and counts["ps_general"] == 0 and live == (True, 0), (counts, live)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    assert np.array_equal(lo, ref["log_obj"]), "log_obj history differs from the oracle"
    assert np.array_equal(ch, ref["chain"]), "chain history differs from the oracle"
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])
    for s in (0, S // 2, S - 1):                           # the monitor's statistic of three slabs against the oracle's
        r = oracle.rhat_gelman(ref["chain"][:, :, s * every:(s + 1) * every])
        assert abs(rmax[s] - np.max(r)) < 1e-9, (s, rmax[s], np.max(r))


def test_c2_tempered_live_run_equals_oracle(demc, oracle):
    """demcz_anneal's accept test log(rand()) < (prop - prev) / T(ig) (demcz_anneal.jl:172-178, T from :1-3) on the steady-state
    wave-per-chain kernel: window_kernel_ps2<.., LIVE, TEMPER> -- the temperatures ride in the pass's DMA -- for 1500 generations
    of 1024 chains in three calls, against the oracle."""
    N, d, K, G, seed = 1024, 5, 10, 1500, 977
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    T = np.array([3.0 * (1e-3 / 3.0) ** (g / G) for g in range(1, G + 1)])
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    for a, b in ((1, 500), (501, 1000), (1001, 1500)):
        e.run(a, b, w["gamma"], T[a - 1:b])
    counts, live = e.kernel_counts(), e.live_status()
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    tot = e.changed_total(1, G)
    e.close()
    assert counts["ps2"] == 3 and counts["ps_general"] == 0 and live == (True, 0), (counts, live)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, temperature=T, threads=THREADS)
    _same(dict(chain=ch, log_obj=lo, X=X, logp=lp, Z=Z, M=M, changed_total=tot), ref)


def test_matrix_form_of_the_wave_per_chain_kernel_equals_oracle(demc, oracle, monkeypatch):
    """window_kernel_pw<0, 20, LIVE, ., MF = true> (DEMCZ_PW_MFMA=1): the 31 candidates of a pass, their whitened residuals and
    their sums of squares as 36 v_mfma_f64_16x16x4_f64 instructions whose accumulation is the oracle's sequential fma chain
    (demcz_kernels_pw.h).  Measured slower than the scalar form and therefore off by default -- but it is
    the same doubles: C4's shard for 2000 generations, plain and tempered, against the oracle."""
    monkeypatch.setenv("DEMCZ_PW_MFMA", "1")
    d, N, K, G, seed = 20, 1024, 10, 2000, 4242
    w = demc.workloads.mvnormal_problem(d, N)
    a = _hip(demc, w, N, d, K, G, [range(d)], seed, w["gamma"], pieces=[700, 5, 1295])
    assert a["lanes"] == SPLIT_WAVE and a["live"] == (True, 0)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _same(a, ref)
    T = np.array([demc.tempbaseline(g, 300, 3.0, 1e-3) for g in range(1, 301)])
    b = _hip(demc, w, N, d, K, 300, [range(d)], seed + 1, w["gamma"], temperature=T)
    refb = oracle_sample(oracle, w["target"], w["Zinit"], N, K, 300, None, w["eps_scale"], w["gamma"], seed + 1, temperature=T, threads=THREADS)
    _same(b, refb)
