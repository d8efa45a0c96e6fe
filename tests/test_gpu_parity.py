"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.
Bit-exact: the arithmetic spec makes every double identical."""
import os

import numpy as np
import pytest

from helpers import SPLIT, SPLIT_WAVE, auto_split_layout, oracle_sample

pytestmark = pytest.mark.gpu


def test_selftest_draws_bit_exact(demc, oracle):
    """Device Philox words, Box-Muller pairs and log(u) == oracle, for 20000 blocks."""
    n, seed, chain, blk0 = 20000, 31953150, 12345, 777
    words, normals, logu = demc.selftest_draws(seed, chain, blk0, n)
    for i in range(0, n, 97):
        r1, r2 = oracle.draw_block(seed, chain, blk0 + i)
        assert (int(words[i, 0]), int(words[i, 1])) == (r1, r2)
        z0, z1 = oracle.normal_pair(r1, r2)
        assert normals[i, 0] == z0 and normals[i, 1] == z1
    u = ((words[:, 0] >> np.uint64(12)).astype(np.float64) + 0.5) * 2.0 ** -52
    assert np.array_equal(logu, oracle.dm_log(u))
    assert abs(normals.mean()) < 0.02 and abs(normals.std() - 1.0) < 0.02


@pytest.mark.parametrize("lanes", [1, 0, "ml", SPLIT, SPLIT_WAVE])
@pytest.mark.parametrize("N,d,G", [(4, 5, 200), (100, 5, 57), (1024, 5, 40), (64, 3, 30), (65, 8, 25), (32, 7, 25),
                                   (13, 2, 31), (77, 4, 33), (50, 10, 27), (41, 20, 23),
                                   # round 5: no whitelist of dimensions for the wave-per-chain consumer (d = 2..32)
                                   (45, 6, 33), (40, 9, 27), (37, 12, 26), (70, 16, 24), (33, 21, 22), (36, 23, 22), (29, 26, 21), (64, 30, 23),
                                   (21, 32, 22)])
def test_mvnormal_full_block_bit_exact(demc, oracle, N, d, G, lanes):
    """Every layout: one lane per chain fused (lanes=1), the library's choice (lanes=0: one wave per chain at every d in 2..32
    for populations this small), eight / sixteen lanes per chain and the split's replicated / cooperating consumers where they
    are built (d <= 10, d = 20), one wave per chain asked for by name."""
    from helpers import split_built
    if lanes in ("ml", SPLIT) and not split_built(d):
        pytest.skip("eight- / sixteen-lane kernels: d <= 10 and d = 20")
    if lanes == "ml":
        lanes = 16 if d == 20 else 8
    w = demc.workloads.mvnormal_problem(d, N)
    seed = 99 + N
    mc, Z, runner = demc.demcz_sample(w["target"], w["Zinit"], N, w["K"], G, 1, [range(d)], w["eps_scale"], w["gamma"],
                                      verbose=False, seed=seed, lanes_per_chain=lanes, return_runner=True)
    used = runner.engines[0].info()["lanes_per_chain"]
    runner.close()
    assert used == (auto_split_layout(d, N) if lanes == 0 else lanes)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, w["K"], G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(mc.chain, ref["chain"])
    assert np.array_equal(mc.log_obj, ref["log_obj"])
    assert np.array_equal(mc.Xcurrent, ref["X"])
    assert np.array_equal(mc.log_objcurrent, ref["logp"])
    assert np.array_equal(Z, ref["Z"])


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


BLOCKS_D6 = [[0], [1, 2], [5, 3, 4]]
BLOCKS_D20 = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]


@pytest.mark.parametrize("N,d,G,blocks", [(70, 6, 35, BLOCKS_D6), (256, 20, 23, BLOCKS_D20), (64, 20, 21, None),
                                          (33, 10, 30, [range(0, 10)]), (40, 5, 30, [[4, 3, 2, 1, 0]]),
                                          (16, 13, 25, [range(0, 7), range(7, 13)]), (8, 64, 12, None),
                                          # consecutive blocks of unequal length: the sums cut at their boundaries
                                          # through the group-start mask -- one-lane, 8- / 16-lane and generic kernels
                                          (48, 10, 30, [range(0, 7), [9, 7, 8]]), (130, 20, 22, [range(0, 10), range(10, 13), range(13, 20)]),
                                          (24, 5, 40, [[1, 0], [2], [4, 3]]), (20, 7, 25, [[0, 1, 2], [3, 4], [6, 5]])])
@pytest.mark.parametrize("lanes", [1, 0])
def test_mvnormal_blocks_and_generic_d_bit_exact(demc, oracle, N, d, G, blocks, lanes):
    """Block updates (C3 layout), permuted single block (not the FULL fast path), runtime-d kernel;
    in the one-lane layout and in the library's choice (8/16 lanes per chain where built)."""
    w = demc.workloads.mvnormal_problem(d, N)
    bl = blocks or [range(d)]
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], N, w["K"], G, len(bl), bl, w["eps_scale"], w["gamma"],
                              verbose=False, seed=5, lanes_per_chain=lanes)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, w["K"], G, [list(b) for b in bl], w["eps_scale"], w["gamma"], 5)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"]) and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("lanes", [1, 0])
@pytest.mark.parametrize("kind,d,N,G", [("iso", 10, 50, 60), ("iso", 7, 20, 30), ("iso", 30, 40, 25), ("linreg", 10, 64, 40), ("linreg", 26, 10, 20),
                                        ("linreg", 4, 10, 20), ("linreg", 10, 37, 33), ("linreg", 10, 5, 12)])
def test_anneal_targets_bit_exact(demc, oracle, kind, d, N, G, lanes):
    """Tempered accept (demcz_anneal.jl:172-178) on the isotropic quadratic and the regression SSE."""
    w = demc.workloads.iso_quad_problem(d, N) if kind == "iso" else demc.workloads.linreg_problem(d, N, nobs=120)
    mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], w["gamma"], verbose=False,
                              T0=3, TN=1e-3, seed=17, adaptγ={"adapt": False}, lanes_per_chain=lanes)
    temps = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], w["gamma"], 17, temperature=temps)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"]) and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("d,N,nobs,lanes", [(26, 40, 1000, 0), (26, 33, 120, 16), (4, 50, 37, 16), (7, 64, 500, 0), (13, 20, 16, 16), (28, 17, 131, 0),
                                            (2, 30, 5, 16), (21, 24, 15, 0), (6, 10, 1600, 0), (9, 7, 1537, 16), (12, 9, 1536, 0)])
def test_regression_target_any_dimension_sixteen_lanes(demc, oracle, d, N, nobs, lanes):
    """Round 5: the regression SSE (test/example_linreg.jl:32) at dimensions other than 10 -- the reference's own example runs
    d = 26 -- on window_kernel_ml<LINREG_SSE, d, 16>: sixteen lanes per chain = the spec's sixteen interleaved partial sums, the
    spec's tree by lane shuffles.  Observation counts that are no multiple of sixteen (and fewer than sixteen), tempered, chosen
    by the library (lanes_per_chain = 0) and by name; bit-exact against the oracle.  Up to 1536 observations the chain wave has
    helper waves (COOP: residuals of a workgroup's four chains in LDS), beyond that it forms the residuals itself: both sides of
    the limit are here."""
    G, K, seed = 40, 10, 23
    w = demc.workloads.linreg_problem(d, N, nobs=nobs)
    temps = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], lanes_per_chain=lanes)
    assert e.info()["lanes_per_chain"] == 16
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, 17, w["gamma"], temps[:17])
    e.run(18, G, w["gamma"], temps[17:])
    assert "window_kernel_ml<LINREG_SSE" in e.kernel_name(), e.kernel_name()
    if nobs > 1536:             # (helper waves only while the residuals fit LDS; the -DML_LRDPP=0 build has none at all)
        assert not e.kernel_name().endswith("false, false, true>"), e.kernel_name()
    elif "flipped" not in os.environ.get("DEMCZ_LIB", ""):
        assert e.kernel_name().endswith("false, false, true>"), e.kernel_name()
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    tot = e.changed_total(1, G)
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, temperature=temps)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])
    assert tot == int(ref["changed"].sum())


def test_anneal_gamma_adaptation_matches_host_logic_on_oracle(demc):
    """Same driver, HIP engine vs the oracle-backed test engine: identical, incl. adapted gamma."""
    from oracle_engine import OracleEngine
    d, N, G = 10, 32, 400
    w = demc.workloads.iso_quad_problem(d, N)
    args = (w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38)
    kw = dict(verbose=False, T0=2, TN=1e-4, seed=6, adaptγ={"adapt_every": 100})
    a, Za = demc.demcz_anneal(*args, **kw)
    b, Zb = demc.demcz_anneal(*args, engine_factory=OracleEngine, **kw)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(Za, Zb)


def test_device_statistics_vs_oracle(demc, oracle):
    """R-hat (utils.jl:2-20), per-chain accept ratio (utils.jl:61), mean/cov (utils.jl:96-111) and the
    per-generation changed counts, computed on the device history."""
    d, N, G = 5, 200, 301
    w = demc.workloads.mvnormal_problem(d, N)
    mc, Z, runner = demc.demcz_sample(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False,
                                      seed=2, return_runner=True)
    e = runner.engines[0]
    for a, b in [(1, G), (2, 300), (100, 299), (290, 301)]:
        tol = dict(rtol=1e-9, atol=0)                               # floating-point reductions: order differs
        assert np.allclose(e.rhat(a, b), oracle.rhat_gelman(mc.chain[:, :, a - 1:b]), **tol)
        assert np.allclose(e.accept_ratio(a, b), oracle.changed_per_chain(mc.log_obj[:, a - 1:b]) / (b - a), rtol=1e-15)
        mean, cov = e.mean_cov(a, b)
        om, oc = oracle.mean_cov_chain(mc.chain[:, :, a - 1:b])
        assert np.allclose(mean, om, rtol=1e-12) and np.allclose(cov, oc, rtol=1e-9, atol=1e-18)
    lp0 = oracle.logp(oracle.Problem(N, d, 10, 10, w["eps_scale"], 2, target=w["target"].spec()), w["Zinit"][-N:])
    prev = np.concatenate([lp0[:, None], mc.log_obj[:, :-1]], axis=1)
    assert np.array_equal(e.get_changed(1, G), (mc.log_obj != prev).sum(axis=0))
    runner.close()


def test_split_calls_and_history_origin(demc, oracle):
    """demcz_run in arbitrary pieces == one call; history origin shifting keeps a sliding window."""
    d, N, G = 5, 96, 83
    w = demc.workloads.mvnormal_problem(d, N)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 12)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * 9, Gcap=30, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=12,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    g = 1
    for step in (3, 7, 10, 1, 9, 30, 23):
        if (g - 1) % 30 + step > 30 or (g - 1) % 30 == 0:
            e.synchronize()
            e.set_history_origin(g - 1)
        e.run(g, g + step - 1, 2.38)
        ch, lo = e.get_history(g, g + step - 1)
        assert np.array_equal(ch, ref["chain"][:, :, g - 1:g + step - 1])
        g += step
    X, lp, Z, M = e.get_state()
    assert M == ref["M"] and np.array_equal(Z, ref["Z"]) and np.array_equal(X, ref["X"])
    with pytest.raises(demc.DemczError) as ei:                       # Mcap exhausted: loud, not silent
        e.run(g, g + 30, 2.38)
    assert ei.value.code in (3,)
    e.close()


def test_host_closure_mode_equals_device_target(demc, oracle):
    """Arbitrary Python closure through demcz_propose / demcz_accept_commit (the reference's
    defining feature, demcz.jl:189): with the closure = the oracle's own log-density the
    trajectory equals the device-target run bit for bit."""
    d, N, G = 6, 24, 25
    w = demc.workloads.mvnormal_problem(d, N)
    # (the oracle's log-density of a run with THESE blocks: its sums are cut at their boundaries)
    prob = oracle.Problem(1, d, 10, 10, w["eps_scale"], 0, blocks=BLOCKS_D6, target=w["target"].spec())
    closure = lambda x: float(oracle.logp(prob, x[None, :])[0])      # noqa: E731
    a, Za = demc.demcz_sample(closure, w["Zinit"], N, 10, G, 3, BLOCKS_D6, w["eps_scale"], 2.38, verbose=False, seed=3)
    b, Zb = demc.demcz_sample(w["target"], w["Zinit"], N, 10, G, 3, BLOCKS_D6, w["eps_scale"], 2.38, verbose=False, seed=3)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(a.log_obj, b.log_obj) and np.array_equal(Za, Zb)


@pytest.mark.parametrize("shards", [2, 4])
def test_in_process_shards_on_gpu_are_invariant(demc, shards):
    """S handles on one GPU, host-driven K-boundary exchange: same bits as one handle."""
    d, N, G = 5, 256, 45
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=20, autostop_Rhat=1.0)
    a, Za, ra = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=8, return_runner=True)
    sh = demc.Sharding(rank=0, world_size=1, mode="host", local_shards=shards)
    b, Zb, rb = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=8, sharding=sh, return_runner=True)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(Za, Zb)
    assert np.allclose(ra.rhat(1, 40), rb.rhat(1, 40), rtol=1e-12)
    assert np.array_equal(ra.changed(1, G), rb.changed(1, G))
    ra.close(); rb.close()


def test_rccl_path_single_rank(demc):
    """demcz_comm_init at nranks=1 (the one-GPU box cannot host two RCCL ranks): communicator
    set-up works and the run is unchanged."""
    d, N, G = 5, 128, 25
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    outs = []
    for use_comm in (False, True):
        e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * 3, Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                           target=w["target"])
        if use_comm:
            e.comm_init(e.comm_unique_id(), 1, 0)
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.run(1, G, 2.38)
        outs.append((e.get_history(1, G)[0], e.rhat(1, G), e.get_state()[2]))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_argument_validation(demc):
    """Error behaviour at the boundary: status codes + messages, never a crash."""
    w = demc.workloads.mvnormal_problem(5, 8)
    with pytest.raises(ValueError):                                  # demcz.jl:176-179 would throw at M=1
        demc.demcz_sample(w["target"], w["Zinit"][:1], 1, 10, 5, verbose=False)
    e = demc.HipEngine(N=8, d=5, K=10, Mcap=100, Gcap=10, blockindex=[range(5)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    with pytest.raises(demc.DemczError) as ei:
        e.run(1, 5, 2.38)                                            # no state yet
    assert ei.value.code == 4
    e.set_state(w["Zinit"][-8:], None, w["Zinit"])
    with pytest.raises(demc.DemczError) as ei:
        e.run(1, 11, 2.38)                                           # beyond the history window
    assert ei.value.code == 3
    with pytest.raises(demc.DemczError):
        e.rhat(1, 3)                                                 # window too short for split-R-hat
    e.close()
    with pytest.raises(demc.DemczError) as ei:
        demc.HipEngine(N=8, d=5, K=10, Mcap=100, Gcap=10, blockindex=[[0, 0]], eps_scale=w["eps_scale"], seed=1, target=w["target"])
    assert ei.value.code == 1


def test_full_size_c2_properties(demc, oracle):
    """BASELINE C2 at full size (N=1024, d=5, 10000 generations): the reference's predicates
    (Rhat, accept band: example_normpdf.jl:49-51) tightened to BASELINE's 1.05, moments of the
    pooled draws against (mu, Sigma), archive bookkeeping, and run-splitting invariance."""
    d, N, G = 5, 1024, 10000
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="no")
    mc, Z, runner = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=31953150, return_runner=True)
    e = runner.engines[0]
    assert Z.shape[0] == w["Zinit"].shape[0] + N * (G // 10)
    assert np.array_equal(Z[w["Zinit"].shape[0]:][:N], mc.chain[:, :, 9])      # rows appended after generation 10
    assert np.array_equal(Z[-N:], mc.chain[:, :, G - 1])
    rh = e.rhat(G - 999, G)
    assert np.max(rh) < 1.05, rh
    acc = e.accept_ratio(G - 2499, G)
    assert np.all(acc > 0.1) and np.all(acc < 0.45), (acc.min(), acc.max())
    mean, cov = e.mean_cov(G // 2 + 1, G)
    sd = np.sqrt(np.diag(w["Sigma"]))
    assert np.all(np.abs(mean - w["mu"]) < 0.01 * sd), (mean - w["mu"]) / sd
    assert np.allclose(cov, w["Sigma"], rtol=0.02, atol=0.01 * w["Sigma"].max())
    # the whole run against the oracle, bit for bit: 10^7 chain-updates, every history entry and every
    # archive row (the launches run through 100 K boundaries each, handing rows from wave to wave:
    # one stale or torn row anywhere would change everything after it)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 31953150)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"])
    assert np.array_equal(Z, ref["Z"]) and np.array_equal(mc.Xcurrent, ref["X"])
    runner.close()


def test_bench_under_torchrun_single_rank(tmp_path):
    """bench.py launched the way the driver launches it (torch.distributed.run, 127.0.0.1): one JSON
    line with the contract's keys.  (N>1 needs N GPUs; the one-GPU box runs the N=1 form.)"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(root / "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "50", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["value"] > 1e6 and d["dtype"] == "f64"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])


@pytest.mark.parametrize("first,second", [(None, "4x5"), ("4x5", None), (None, "10-3-7")])
def test_prevrun_carried_into_a_run_with_other_blocks_equals_oracle(demc, oracle, first, second):
    """demcz.jl:18-22: a `prevrun` hands X AND log_objcurrent to the next run.  The MvNormal sums are cut at the run's block
    boundaries, so log-densities made under Nblocks = 1 are not, to the last bit, what a run with four
    blocks would compute at the same points (tests/test_oracle_targets.py: within 8 ulp) -- and the next run must still start
    from the CARRIED values, as the reference does, not from its own re-evaluation: its first accept tests compare a grouped
    proposal with an ungrouped current value.  The block-update kernel with the incremental quadratic form (window_kernel_mlb,
    QB = 5) keeps per-chain partial sums: they must come from X while log p comes from the caller.  HIP path = oracle doing the
    same, bit for bit, in both directions and for unequal blocks (the full-evaluation form)."""
    d, N, K, G1, G2, seed = 20, 256, 10, 60, 70, 77
    layouts = {None: None, "4x5": [list(range(0, 5)), list(range(5, 10)), list(range(10, 15)), list(range(15, 20))],
               "10-3-7": [list(range(0, 10)), list(range(10, 13)), list(range(13, 20))]}
    b1, b2 = layouts[first], layouts[second]
    w = demc.workloads.mvnormal_problem(d, N)
    bl = lambda b: b or [range(d)]
    from demc_jl_amd.sampler import blocks_per_generation
    S1, S2 = blocks_per_generation(bl(b1)), blocks_per_generation(bl(b2))
    assert S1 != S2       # (generations of different size: the resumed streams start behind everything run 1 drew, never inside it)
    mc1, Z1 = demc.demcz_sample(w["target"], w["Zinit"], N, K, G1, len(bl(b1)), bl(b1), w["eps_scale"], w["gamma"], verbose=False, seed=seed)
    mc2, Z2 = demc.demcz_sample(w["target"], Z1, N, K, G2, len(bl(b2)), bl(b2), w["eps_scale"], w["gamma"], verbose=False, seed=seed,
                                prevrun=mc1)
    r1 = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G1, b1, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(mc1.chain, r1["chain"]) and np.array_equal(mc1.log_objcurrent, r1["logp"]) and np.array_equal(Z1, r1["Z"])
    # the second run: state and log-densities carried (NOT recomputed under the new blocks), streams advanced by G1 generations
    r2 = oracle_sample(oracle, w["target"], r1["Z"], N, K, G2, b2, w["eps_scale"], w["gamma"], seed, X0=r1["X"], lp0=r1["logp"],
                       rng_offset=-(-G1 * S1 // S2))
    assert np.array_equal(mc2.chain[:, :, G1:], r2["chain"]), "the resumed run differs from the oracle's"
    assert np.array_equal(mc2.log_obj[:, G1:], r2["log_obj"]) and np.array_equal(Z2, r2["Z"])
    own = oracle.logp(r2["prob"], r1["X"])
    assert not np.array_equal(own, r1["logp"]), "the case must exercise carried values that differ from the run's own evaluation"


def test_utils_mirror_reference_postprocessing(demc, oracle, tmp_path):
    """The post-processing of test/example_normpdf.jl:35-51 with the reference's function names,
    reduced on the GPU from host arrays; plus resume (prevrun) and the checkpoint file."""
    d, N, G = 5, 5, 2000
    w = demc.workloads.mvnormal_problem(d, N)
    Z0 = w["Zinit"][:50]
    args = (N, 10, G, 1, [range(d)], w["eps_scale"], 2.38)
    mc, Z = demc.demcz_sample(w["target"], Z0, *args, verbose=False, seed=7)
    mc2, Z2 = demc.demcz_sample(w["target"], Z[-51:], *args, prevrun=mc, verbose=False, seed=7)   # example_normpdf.jl:32
    assert mc2.chain.shape == (N, d, 2 * G)
    keep = slice(2 * G - G // 2, 2 * G)                              # :35-39
    chain_burned, logobj_burned = mc2.chain[:, :, keep], mc2.log_obj[:, keep]
    flat = demc.flatten_chain(chain_burned, N, G // 2, d)           # :40
    assert flat.shape == (d, N * (G // 2)) and flat[2, 3 * N + 1] == chain_burned[1, 2, 3]   # utils.jl:26-29 ordering
    b, Sb = demc.mean_cov_chain(chain_burned, N, G // 2, d)         # :44
    ob, oS = oracle.mean_cov_chain(chain_burned)
    assert np.allclose(b, ob, rtol=1e-12) and np.allclose(Sb, oS, rtol=1e-9, atol=1e-18)
    acc, Rhat = demc.convergence_check(chain_burned, logobj_burned, "none", verbose=False)   # :47
    assert np.allclose(Rhat, oracle.rhat_gelman(chain_burned), rtol=1e-9)
    assert np.allclose(acc, oracle.changed_per_chain(logobj_burned) / (G // 2 - 1), rtol=1e-15)
    assert np.allclose(demc.Rhat_gelman(chain_burned, N, G // 2, d), Rhat)
    # resume == uninterrupted run (K | G), through the checkpoint file
    one, Zone = demc.demcz_sample(w["target"], Z0, N, 10, 2 * G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False, seed=7)
    demc.save_checkpoint(tmp_path / "ck.npz", mc, Z, G, 7)
    prev, Zc, done, seed = demc.load_checkpoint(tmp_path / "ck.npz")
    # (no rng_offset: the checkpoint's prevrun itself records how far the streams have advanced)
    res, Zres = demc.demcz_sample(w["target"], Zc, *args, prevrun=prev, verbose=False, seed=seed)
    assert done == G and prev.generations_drawn == G and res.generations_drawn == 2 * G
    assert np.array_equal(res.chain[:, :, 1:], one.chain[:, :, G:]) and np.array_equal(Zres, Zone)


@pytest.mark.parametrize("case", ["d1", "K1", "N1", "M0_2", "nobs5", "nobs17", "eps0", "single_param_blocks"])
def test_edge_cases_bit_exact(demc, oracle, case):
    """Smallest / degenerate shapes: d = 1 (unscaled gamma and ONE scalar normal, demcz.jl:183-184),
    K = 1 (append after every generation), one chain, two archive rows (the minimum for two distinct
    draws, demcz.jl:176-179), fewer observations than SSE partials, zero jitter, all blocks of size 1."""
    kw = dict(verbose=False, seed=77)
    if case == "d1":
        w = demc.workloads.mvnormal_problem(1, 9); N, K, G, blocks = 9, 10, 40, [range(1)]
    elif case == "K1":
        w = demc.workloads.mvnormal_problem(5, 12); N, K, G, blocks = 12, 1, 25, [range(5)]
    elif case == "N1":
        w = demc.workloads.mvnormal_problem(3, 1); N, K, G, blocks = 1, 10, 60, [range(3)]
    elif case == "M0_2":
        w = demc.workloads.mvnormal_problem(4, 2); w["Zinit"] = np.asfortranarray(w["Zinit"][:2]); N, K, G, blocks = 2, 10, 35, [range(4)]
    elif case in ("nobs5", "nobs17"):
        w = demc.workloads.linreg_problem(10, 8, nobs=5 if case == "nobs5" else 17); N, K, G, blocks = 8, 10, 30, [range(10)]
    elif case == "eps0":
        w = demc.workloads.mvnormal_problem(5, 16); w["eps_scale"] = np.zeros(5); N, K, G, blocks = 16, 10, 30, [range(5)]
    else:
        w = demc.workloads.mvnormal_problem(5, 10); N, K, G, blocks = 10, 10, 30, [[3], [0], [4], [1], [2]]
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], N, K, G, len(blocks), blocks, w["eps_scale"], w["gamma"], **kw)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, [list(b) for b in blocks], w["eps_scale"], w["gamma"], 77)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"]) and np.array_equal(Z, ref["Z"])
    assert Z.shape[0] == w["Zinit"].shape[0] + N * (G // K)


def test_no_history_handle_and_closure_anneal(demc, oracle):
    """Gcap = 0 (states and archive only, e.g. burn-in): same final state; and an annealed Python
    closure through the host-closure path equals the device target."""
    d, N, G = 5, 40, 37
    w = demc.workloads.mvnormal_problem(d, N)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 4)
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=w["Zinit"].shape[0] + N * 4, Gcap=0, blockindex=[range(d)], eps_scale=w["eps_scale"],
                       seed=4, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G, 2.38)
    X, lp, Z, M = e.get_state()
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and np.array_equal(Z, ref["Z"])
    with pytest.raises(demc.DemczError):
        e.get_history(1, G)
    e.close()
    w = demc.workloads.iso_quad_problem(10, 12)
    prob = oracle.Problem(1, 10, 10, 10, w["eps_scale"], 0, target=w["target"].spec())
    closure = lambda x: float(oracle.logp(prob, x[None, :])[0])      # noqa: E731
    args = (w["Zinit"], 12, 10, 30, 1, [range(10)], w["eps_scale"], 2.38)
    kw = dict(verbose=False, T0=2, TN=1e-2, seed=8, adaptγ={"adapt_every": 10})
    a, Za = demc.demcz_anneal(closure, *args, **kw)
    b, Zb = demc.demcz_anneal(w["target"], *args, **kw)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(Za, Zb)


def _engine_run(demc, w, N, d, G, blocks, seed, pieces=None, temperature=None, gamma=None):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    g = 1
    for n in (pieces or [G]):
        e.run(g, g + n - 1, w["gamma"] if gamma is None else gamma, None if temperature is None else temperature[g - 1:g + n - 1])
        g += n
    assert g == G + 1
    return e


def test_full_size_c3_properties(demc, oracle):
    """BASELINE C3 at full width (d=20 in four blocks of five, N=4096), 4000 generations on the device:
    first generations against the oracle bit for bit, the run cut into uneven pieces == one call, archive
    bookkeeping, and the accept band / moments of the reference's predicates (example_normpdf.jl:42-51) once
    the N(0,1) start has been forgotten (about 2500 generations at d=20: scripts/c3_stats.py)."""
    d, N, G = 20, 4096, 4000
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine_run(demc, w, N, d, G, BLOCKS_D20, 31953150)
    ch, lo = e.get_history(1, 12)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, 12, [list(b) for b in BLOCKS_D20], w["eps_scale"], w["gamma"], 31953150)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    X, lp, Z, M = e.get_state()
    assert M == w["Zinit"].shape[0] + N * (G // 10)
    last, _ = e.get_history(G, G)
    assert np.array_equal(Z[-N:], last[:, :, 0]) and np.array_equal(X, last[:, :, 0])
    acc = e.accept_ratio(G - 999, G)
    assert np.all(acc > 0.1) and np.all(acc < 0.6), (acc.min(), acc.max())
    mean, cov = e.mean_cov(G - 999, G)
    sd = np.sqrt(np.diag(w["Sigma"]))
    assert np.all(np.abs(mean - w["mu"]) < 0.05 * sd), np.abs((mean - w["mu"]) / sd).max()
    assert abs(np.trace(cov) / np.trace(w["Sigma"]) - 1) < 0.05
    rh = e.rhat(G - 999, G)
    assert np.all(np.isfinite(rh)) and rh.max() < 1.5, rh
    e.close()
    e2 = _engine_run(demc, w, N, d, G, BLOCKS_D20, 31953150, pieces=[7, 993, 1, 409, 2590])
    X2, lp2, Z2, M2 = e2.get_state()
    e2.close()
    assert M2 == M and np.array_equal(X2, X) and np.array_equal(lp2, lp) and np.array_equal(Z2, Z)


def test_full_size_c4_shards_are_invariant(demc, oracle):
    """BASELINE C4's shape (d=20, N=8192 as 8 shards of 1024 chains, replicated archive, R-hat autostop
    statistics reduced across shards): eight handles on the one GPU with the host-driven K-boundary
    exchange give the bits of one 8192-chain handle, and both start like the oracle."""
    d, N, G = 20, 8192, 300
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=100, autostop_Rhat=1.0)
    a, Za, ra = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=31953150, return_runner=True)
    sh = demc.Sharding(rank=0, world_size=1, mode="host", local_shards=8)
    b, Zb, rb = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=31953150, sharding=sh, return_runner=True)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(a.log_obj, b.log_obj) and np.array_equal(Za, Zb)
    assert np.allclose(ra.rhat(101, 300), rb.rhat(101, 300), rtol=1e-11)
    assert np.array_equal(ra.changed(1, G), rb.changed(1, G))
    ra.close(); rb.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, 12, None, w["eps_scale"], w["gamma"], 31953150)
    assert np.array_equal(a.chain[:, :, :12], ref["chain"])


def test_full_size_c5_properties(demc, oracle):
    """BASELINE C5 at full size (regression SSE, d=10, nobs=1000, N=2048, T0=3 -> TN=1e-3 over 10000
    generations, gamma adaptation on): the first generations against the oracle bit for bit; at the end
    of the schedule the population sits on the least-squares solution (test/example_linreg.jl:59-66 compares
    against OLS by eye; here it is asserted) and log_obj never exceeds the optimum."""
    d, N, G = 10, 2048, 10000
    w = demc.workloads.linreg_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], γ=w["gamma"], verbose=False, T0=3, TN=1e-3,
                        autostop="no")
    mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], opts, seed=319531501)
    beta_ols, *_ = np.linalg.lstsq(w["design"], w["y"], rcond=None)
    best = -0.5 * np.sum((w["y"] - w["design"] @ beta_ols) ** 2)
    assert np.all(mc.log_objcurrent <= best * (1 - 1e-12))
    assert np.all(mc.log_objcurrent > best - 0.05), (best - mc.log_objcurrent).max()
    assert np.max(np.abs(mc.Xcurrent - beta_ols)) < 0.02
    assert Z.shape[0] == w["Zinit"].shape[0] + N * (G // 10)
    G0 = 12
    temps = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G0 + 1)])
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G0, None, w["eps_scale"], w["gamma"], 319531501, temperature=temps)
    assert np.array_equal(mc.chain[:, :, :G0], ref["chain"]) and np.array_equal(mc.log_obj[:, :G0], ref["log_obj"])


@pytest.mark.parametrize("layout", [SPLIT, SPLIT_WAVE])
def test_live_handoff_under_uneven_load(demc, oracle, layout):
    """The in-launch row hand-off of the split layout while another stream keeps the
    chip busy with a 2^18-chain population: consumer waves are dispatched late and unevenly, rows arrive
    late -- the result must still be the oracle's, bit for bit, for K = 10, 3 and 1 (both consumers)."""
    d, N = 5, 1024
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    Nb = 1 << 18
    wb = demc.workloads.mvnormal_problem(d, Nb)
    big = demc.HipEngine(N=Nb, d=d, K=10, Mcap=wb["Zinit"].shape[0] + Nb * 31, Gcap=0, blockindex=[range(d)],
                         eps_scale=wb["eps_scale"], seed=3, target=wb["target"])
    big.set_state(wb["Zinit"][-Nb:], None, wb["Zinit"])
    for K, G in ((10, 600), (3, 200), (1, 60)):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                           seed=11, target=w["target"], lanes_per_chain=layout)
        assert e.info()["lanes_per_chain"] == layout
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        g = 1
        for piece in (G // 3, G // 3, G - 2 * (G // 3)):
            _big_step(big)
            e.run(g, g + piece - 1, 2.38)         # asynchronous: overlaps the big population's windows on the other stream
            g += piece
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        e.close()
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], 2.38, 11)
        assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"]) and np.array_equal(Z, ref["Z"]), K
    big.synchronize()
    big.close()


_BIG_G = [1]


def _big_step(big):
    """30 more generations (3 windows, ~0.1 ms each at 2^18 chains) of the background population, not waited for."""
    g = _BIG_G[0]
    big.run(g, g + 29, 2.38)
    _BIG_G[0] = g + 30


def test_run_checked_is_the_driver_loop(demc, oracle):
    """demcz_run_checked (demcz.jl:30-55 as one call) == the host loop over demcz_run + demcz_rhat: same
    stop generation, same R-hat values, same chains; the window kernels' event timing counts the launches."""
    d, N, G, every = 5, 256, 600, 100
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]

    def engine():
        e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                           seed=4, target=w["target"])
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        return e

    a = engine()
    trace_a, g = [], 1
    while g <= G:
        a.run(g, g + every - 1, 2.38)
        trace_a.append(a.rhat(g, g + every - 1))
        g += every
    b = engine()
    b.set_kernel_timing(True)
    g_stop, mx, last = b.run_checked(1, G, 2.38, every, 0.0)
    n, ms = b.get_kernel_time()
    assert g_stop == G and n >= 1 and ms > 0.0
    assert np.array_equal(mx, [r.max() for r in trace_a]) and np.array_equal(last, trace_a[-1])
    assert np.array_equal(a.get_history(1, G)[0], b.get_history(1, G)[0])
    # with a threshold: stops at the first check below it, nothing after it has run
    thr = float(np.sort(mx)[len(mx) // 2]) * (1 + 1e-12)
    c = engine()
    g_stop, mx_c, _ = c.run_checked(1, G, 2.38, every, thr)
    k = int(np.argmax(mx < thr))
    assert g_stop == (k + 1) * every and len(mx_c) == k + 1 and c.M == M0 + N * (g_stop // 10)
    # (the library ran the slab after the stop ahead of the decision and discarded it:) the state is that of a run
    # that ended at g_stop, and the run continues from it like any other
    d2 = engine()
    d2.run(1, g_stop, 2.38)
    for x, y in zip(c.get_state(), d2.get_state()):
        assert np.array_equal(x, y)
    if g_stop + 37 <= G:
        c.run(g_stop + 1, g_stop + 37, 2.38)
        d2.run(g_stop + 1, g_stop + 37, 2.38)
        for x, y in zip(c.get_state(), d2.get_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(c.get_history(1, g_stop + 37)[0], d2.get_history(1, g_stop + 37)[0])
    for e in (a, b, c, d2):
        e.close()
    # the sampler's autostop path goes through it and returns what the reference's loop would
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=every, autostop_Rhat=thr)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=4)
    assert mc.chain.shape[2] == (k + 1) * every and Z.shape[0] == M0 + N * ((k + 1) * every // 10)


def test_randomised_shapes_against_the_oracle(demc, oracle):
    """Seeded random shapes through the library's own layout choice (split layouts with in-launch hand-off for the
    dimensions they are built for): odd chain counts, K from 1 to 15, block structures, calls cut at arbitrary
    generations -- chain, log_obj and the archive bit for bit."""
    rng = np.random.default_rng(20240607)
    for case in range(40):
        d = int(rng.choice([2, 3, 4, 5, 8, 10, 20, 6, 7]))
        N = int(rng.integers(1, 200))
        K = int(rng.integers(1, 16))
        G = int(rng.integers(5, 130))
        blocked = d in (5, 6, 10, 20) and rng.random() < 0.3
        if blocked:
            cuts = sorted(set(int(c) for c in rng.integers(1, d, size=int(rng.integers(1, 4)))))
            edges = [0] + cuts + [d]
            perm = rng.permutation(d) if rng.random() < 0.5 else np.arange(d)
            blocks = [[int(p) for p in perm[a:b]] for a, b in zip(edges[:-1], edges[1:])]
        else:
            blocks = [list(range(d))]
        w = demc.workloads.mvnormal_problem(d, max(N, 2))
        M0 = w["Zinit"].shape[0]
        seed = int(rng.integers(1, 2**31))
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=seed,
                           target=w["target"])
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        g = 1
        while g <= G:
            step = int(min(G - g + 1, rng.integers(1, 60)))
            e.run(g, g + step - 1, w["gamma"])
            g += step
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        lanes = e.info()["lanes_per_chain"]
        e.close()
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, blocks, w["eps_scale"], w["gamma"], seed)
        tag = (case, d, N, K, G, blocks, lanes)
        assert np.array_equal(ch, ref["chain"]), tag
        assert np.array_equal(lo, ref["log_obj"]) and np.array_equal(Z, ref["Z"]) and np.array_equal(X, ref["X"]), tag


def test_randomised_annealed_targets_against_the_oracle(demc, oracle):
    """The same for the tempered accept on the isotropic quadratic and the regression SSE (matrix-core residuals,
    split form), with random temperatures, observation counts that are not multiples of 16, and cut calls."""
    rng = np.random.default_rng(77)
    for case in range(16):
        kind = "iso" if rng.random() < 0.4 else "linreg"
        N = int(rng.integers(1, 90))
        K = int(rng.integers(1, 13))
        G = int(rng.integers(5, 70))
        d = 10
        if kind == "iso":
            w = demc.workloads.iso_quad_problem(d, max(N, 2))
        else:
            w = demc.workloads.linreg_problem(d, max(N, 2), nobs=int(rng.integers(3, 150)))
        M0 = w["Zinit"].shape[0]
        seed = int(rng.integers(1, 2**31))
        temps = np.exp(rng.uniform(-6, 1, size=G))
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                           target=w["target"])
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        g = 1
        while g <= G:
            step = int(min(G - g + 1, rng.integers(1, 40)))
            e.run(g, g + step - 1, w["gamma"], temps[g - 1:g + step - 1])
            g += step
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        lanes = e.info()["lanes_per_chain"]
        e.close()
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, temperature=temps)
        tag = (case, kind, N, K, G, lanes)
        assert np.array_equal(ch, ref["chain"]), tag
        assert np.array_equal(lo, ref["log_obj"]) and np.array_equal(Z, ref["Z"]) and np.array_equal(X, ref["X"]), tag


# ---- temperature == 0 (demcz_anneal.jl:18 default TN = 0.; test/test_anneal_parallel.jl uses it) -----------------
def _temps_with_zeros(G):
    """T(ig) = tempbaseline(ig, G, 5, 0.) is 0 for every ig >= 1 (demcz_anneal.jl:1-3): (lp' - lp)/0 = +-Inf accepts
    every improvement and nothing else, 0/0 = NaN rejects (demcz_anneal.jl:172-178).  A few finite and denormal-range
    temperatures are mixed in so the tempered division itself is exercised next to the zeros."""
    T = np.zeros(G)
    T[::7] = 2.5
    T[3::11] = 1e-300
    T[5::13] = 1e300
    return T


@pytest.mark.parametrize("kind,d,N,blocks,lanes", [
    ("iso", 10, 48, None, 0),            # split layout, replicated consumer, TEMPER instantiation
    ("iso", 10, 48, None, 8),            # 8 lanes per chain, fused
    ("iso", 10, 48, None, 1),            # one lane per chain
    ("mvn", 5, 70, None, 0),             # split layout, MvNormal, tempered
    ("mvn", 20, 24, None, 0),            # split form of the 16-lane layout
    ("mvn", 20, 24, None, 16),           # 16 lanes fused
    ("mvn", 6, 30, [[0], [1, 2], [5, 3, 4]], 0),     # block updates, split form
    ("mvn", 6, 30, [[0], [1, 2], [5, 3, 4]], 8),     # block updates, fused
    ("mvn", 6, 30, [[0], [1, 2], [5, 3, 4]], 1),     # block updates, one lane
    ("mvn", 7, 20, None, 1),             # runtime-d kernel
    ("linreg", 10, 40, None, 0),         # regression target, split form (matrix-core residuals)
    ("linreg", 10, 40, None, 16),        # regression target, fused
    ("linreg", 10, 40, None, 1),
])
def test_zero_temperature_bit_exact(demc, oracle, kind, d, N, blocks, lanes):
    G, K, seed = 45, 10, 23
    w = (demc.workloads.iso_quad_problem(d, N) if kind == "iso" else
         demc.workloads.linreg_problem(d, N, nobs=70) if kind == "linreg" else demc.workloads.mvnormal_problem(d, N))
    bl = blocks or [range(d)]
    T = _temps_with_zeros(G)
    mc, Z, = demc.demcz_anneal(w["target"], w["Zinit"], N, K, G, len(bl), bl, w["eps_scale"], w["gamma"], verbose=False,
                               seed=seed, adaptγ={"adapt": False}, lanes_per_chain=lanes,
                               temperaturefun=lambda ig, Ng, T0, TN: float(T[ig - 1]))
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, [list(b) for b in bl] if blocks else None, w["eps_scale"],
                        w["gamma"], seed, temperature=T)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"]) and np.array_equal(Z, ref["Z"])
    # at T = 0 the log-density never decreases (greedy), and generations with T = 0 did accept improvements
    zero = np.flatnonzero(T == 0.0)
    zero = zero[zero > 0]
    assert np.all(mc.log_obj[:, zero] >= mc.log_obj[:, zero - 1])
    assert np.any(mc.log_obj[:, zero] > mc.log_obj[:, zero - 1])


def test_reference_default_TN_zero_schedule(demc, oracle):
    """demcz_anneal's positional default TN = 0. (demcz_anneal.jl:18) through the real schedule function."""
    d, N, G = 10, 16, 60
    w = demc.workloads.iso_quad_problem(d, N)
    mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False, T0=5.0,
                              seed=3, adaptγ={"adapt": False})            # TN defaults to 0.
    T = np.array([demc.tempbaseline(g, G, 5.0, 0.0) for g in range(1, G + 1)])
    assert not T.any()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 3, temperature=T)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("tempered", [False, True])
def test_host_closure_path_equals_oracle(demc, oracle, tempered):
    """The host-closure mode (demcz_propose / demcz_accept_commit / demcz_end_generation: update_demcz_chain_block cut
    at the closure call demcz.jl:189) against the ORACLE directly, block updates and zero temperatures included.
    The closure is the oracle's own log-density, so every double must agree."""
    d, N, G, K, seed = 6, 21, 25, 5, 77
    w = demc.workloads.mvnormal_problem(d, N)
    blocks = [[0], [1, 2], [5, 3, 4]]
    prob1 = oracle.Problem(1, d, K, 10, w["eps_scale"], 0, blocks=blocks, target=w["target"].spec())      # (same blocks: same summation order)
    closure = lambda x: float(oracle.logp(prob1, np.asarray(x)[None, :])[0])
    T = _temps_with_zeros(G) if tempered else None
    if tempered:
        mc, Z = demc.demcz_anneal(closure, w["Zinit"], N, K, G, len(blocks), blocks, w["eps_scale"], w["gamma"], verbose=False,
                                  seed=seed, adaptγ={"adapt": False}, temperaturefun=lambda ig, Ng, T0, TN: float(T[ig - 1]))
    else:
        mc, Z = demc.demcz_sample(closure, w["Zinit"], N, K, G, len(blocks), blocks, w["eps_scale"], w["gamma"], verbose=False,
                                  seed=seed)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, blocks, w["eps_scale"], w["gamma"], seed, temperature=T)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"])
    assert np.array_equal(mc.Xcurrent, ref["X"]) and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("mode", ["copies", "pinned", "pinned-own-pointers"])
def test_host_closure_round_trips_equal_oracle(demc, oracle, mode):
    """The three ways through demcz_propose / demcz_accept_commit: the copying calls of rounds 1-4, the pinned buffers the kernels
    address directly (round 5: demcz_closure_buffers -- no copies, the host spins on the propose kernel's flag, the commit is only
    enqueued), and the pinned buffers with the caller's own arrays passed all the same.  A batched closure (one call per
    block-step), tempered, blocks of unequal length; against the oracle, bit for bit."""
    d, N, G, K, seed = 6, 300, 40, 5, 11
    w = demc.workloads.mvnormal_problem(d, N)
    blocks = [[0, 1], [2], [5, 3, 4]]
    prob = oracle.Problem(N, d, K, 10, w["eps_scale"], 0, blocks=blocks, target=w["target"].spec())
    closure = lambda X: oracle.logp(prob, np.asarray(X))
    T = _temps_with_zeros(G)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=seed,
                       target=closure)
    X0 = np.asfortranarray(w["Zinit"][-N:])
    e.set_state(X0, closure(X0), w["Zinit"])
    if mode != "copies":
        bx, bl = e.closure_buffers()
    for g in range(1, G + 1):
        for ib in range(len(blocks)):
            if mode == "pinned":
                Xp = e.propose(g, ib, w["gamma"])
                assert Xp is bx
                bl[:] = closure(Xp)
                e.accept_commit(None, float(T[g - 1]))
            elif mode == "copies":
                Xp = e.propose(g, ib, w["gamma"])
                e.accept_commit(closure(Xp), float(T[g - 1]))
            else:
                own = np.empty((N, d), order="F")
                e._chk(e._L.demcz_propose(e._h, g, ib, float(w["gamma"]), own.ctypes.data_as(demc._lib._dp)))
                assert np.array_equal(own, bx)
                e.accept_commit(closure(own).copy(), float(T[g - 1]))
        e.end_generation(g)
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, blocks, w["eps_scale"], w["gamma"], seed, temperature=T)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and np.array_equal(Z, ref["Z"])


def test_batched_closure_through_the_sampler_surface(demc, oracle):
    """demcz_sample with a closure that declares `batched = True`: one call per block-step with the N x d proposals."""
    d, N, G, K, seed = 5, 128, 30, 10, 3
    w = demc.workloads.mvnormal_problem(d, N)
    prob = oracle.Problem(N, d, K, 10, w["eps_scale"], 0, target=w["target"].spec())
    calls = []

    def closure(X):
        X = np.asarray(X)
        calls.append(X.shape)
        return oracle.logp(prob, X if X.ndim == 2 else X[None, :])
    closure.batched = True
    mc, Z = demc.demcz_sample(closure, w["Zinit"], N, K, G, 1, [range(d)], w["eps_scale"], w["gamma"], verbose=False, seed=seed)
    assert calls.count((N, d)) >= G
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(Z, ref["Z"])


def test_streamed_history_equals_copied_history(demc, oracle):
    """demcz_history_stream: the pinned host mirrors filled slab by slab while the GPU runs hold exactly what demcz_get_history
    copies afterwards -- through LIVE launches, a forced hand-off redo (the redone slabs are streamed again), a discarded
    speculative slab (zeros), and the drop-in surface (demcz_sample returns arrays over the mirrors)."""
    d, N, K, G = 5, 512, 10, 600
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    for force_redo in (False, True):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=9,
                           target=w["target"])
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.history_stream(True)
        if force_redo:
            e.debug_set_live_fault(1, 301)              # poll limit 1 from generation 301 on: the second call's hand-off fails
        e.run(1, 300, w["gamma"])
        e.run(301, G, w["gamma"])
        ch_copy, lo_copy = e.get_history(1, G)
        ch, lo = e.take_history(1, G)
        assert e.live_status()[1] == (1 if force_redo else 0)
        e.close()
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], 9)
        assert np.array_equal(ch, ch_copy) and np.array_equal(lo, lo_copy)
        assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
        assert ch.flags.f_contiguous and lo.flags.f_contiguous
        del ch, lo                                      # (the mirrors go back to the library's pool)
    # autostop with a threshold: the slab that ran ahead of the stop decision is discarded -- zeros in the mirrors too
    opts = demc.demcopt(d, N=N, K=K, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat", autostop_every=100, autostop_Rhat=1.3)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=9)
    g_stop = mc.chain.shape[2]
    assert g_stop < G and g_stop % 100 == 0
    assert np.array_equal(mc.chain, ref["chain"][:, :, :g_stop]) and np.array_equal(mc.log_obj, ref["log_obj"][:, :g_stop])


def test_bench_sharded_selfcheck_with_a_one_rank_communicator(demc):
    """bench.py's gate in front of every multi-GPU number (sharded_selfcheck): a small sharded run against the same run on one
    GPU, both schedules.  The box has one GPU, so the communicator has one rank -- the RCCL calls, the batching and the
    comparison are the ones an 8-rank run makes."""
    import importlib.util
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_mod", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from demc_jl_amd.sampler import Sharding
    made = []

    def factory(**kw):
        e = demc.HipEngine(**kw)
        e.comm_init(e.comm_unique_id(), 1, 0)          # the library's sharded path: all-gather, scatter, all-reduced R-hat
        made.append(e)
        return e
    sh = Sharding(rank=0, world_size=1, mode="rccl", local_shards=1, all_gather=lambda a: [a], all_reduce_sum=lambda a: a,
                  broadcast_bytes=lambda b: b)
    out = bench.sharded_selfcheck(demc, sh, 0, 0, None, lambda v: v, sharded_engine_factory=factory)
    assert len(made) == 2                              # one sharded engine per schedule
    assert out == {"every_K": True, "batches_of_2": True}, out
