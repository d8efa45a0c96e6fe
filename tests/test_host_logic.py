"""Host-side mirror of the reference's driver surface, exercised on CPU with the oracle-backed
test engine (tests/oracle_engine.py) injected in place of HipEngine."""
import numpy as np
import pytest

import demc_jl_amd as demc
from helpers import oracle_sample
from oracle_engine import OracleEngine


def test_demcopt_defaults_match_reference():
    """src/DEMC.jl:41"""
    o = demc.demcopt(7)
    assert (o.N, o.K, o.Ngeneration, o.Nblocks) == (4, 10, 5000, 1)
    assert [list(b) for b in o.blockindex] == [list(range(7))]
    assert np.array_equal(o.eps_scale, 1e-4 * np.ones(7))
    assert o.γ == 2.38 and o.gamma == 2.38 and o.verbose is True and o.print_step == 100
    assert (o.T0, o.TN, o.autostop, o.autostop_every, o.autostop_Rhat) == (3.0, 1e-3, "Rhat", 1000, 1.05)
    o.γ = 2.0
    o.N = 5                                                          # users mutate fields: example_linreg.jl:37-50
    assert o.gamma == 2.0
    assert demc.demcopt(3, autostop=":no").autostop == "no"


def test_positional_matches_oracle_and_result_shapes(oracle):
    d, N, G = 5, 8, 35
    w = demc.workloads.mvnormal_problem(d, N)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False,
                              seed=3, engine_factory=OracleEngine)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 3)
    assert mc.chain.shape == (N, d, G) and mc.log_obj.shape == (N, G)
    assert mc.Xcurrent.shape == (N, d) and mc.log_objcurrent.shape == (N,)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(Z, ref["Z"])
    assert np.array_equal(mc.Xcurrent, mc.chain[:, :, -1]) and np.array_equal(mc.log_objcurrent, mc.log_obj[:, -1])
    assert Z.shape[0] == w["Zinit"].shape[0] + N * (G // 10)        # demcz.jl:88-91
    _, Zp = demc.demcz_sample(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False,
                              seed=3, engine_factory=OracleEngine, padded_Z=True)
    assert Zp.shape[0] == w["Zinit"].shape[0] + -(-N * G // 10) and np.all(Zp[Z.shape[0]:] == 0)   # demcz.jl:11


def test_init_modes(oracle):
    d, N = 5, 4
    w = demc.workloads.mvnormal_problem(d, N)
    mc, _ = demc.demcz_sample(w["target"], w["Zinit"], N, 10, 20, verbose=False, eps_scale=w["eps_scale"],
                              engine_factory=OracleEngine, init="reference_zeros")
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, 20, None, w["eps_scale"], 2.38, 0, init="zeros")
    assert np.array_equal(mc.chain, ref["chain"])                    # Q1: the serial driver starts at zeros
    # ... but only when the zero padding is at least N rows long (demcz.jl:11,15): 1 generation pads 1 row
    mc1, _ = demc.demcz_sample(w["target"], w["Zinit"], N, 10, 1, verbose=False, eps_scale=w["eps_scale"],
                               engine_factory=OracleEngine, init="reference_zeros")
    X0 = np.vstack([w["Zinit"][-3:], np.zeros((1, d))])
    ref1 = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, 1, None, w["eps_scale"], 2.38, 0, X0=X0)
    assert np.array_equal(mc1.chain, ref1["chain"])
    with pytest.raises(ValueError):
        demc.demcz_sample(w["target"], w["Zinit"][:3], N, 10, 5, verbose=False, engine_factory=OracleEngine)


def test_prevrun_concatenates_and_resumes(oracle):
    """demcz.jl:18-22, 58-59 and the usage of test/example_normpdf.jl:30-32."""
    d, N, G = 5, 5, 40
    w = demc.workloads.mvnormal_problem(d, N)
    kw = dict(verbose=False, engine_factory=OracleEngine)
    mc1, Z1 = demc.demcz_sample(w["target"], w["Zinit"][:50], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, seed=1, **kw)
    mc2, Z2 = demc.demcz_sample(w["target"], Z1[-51:], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, seed=2,
                                prevrun=mc1, **kw)
    assert mc2.chain.shape == (N, d, 2 * G) and mc2.log_obj.shape == (N, 2 * G)
    assert np.array_equal(mc2.chain[:, :, :G], mc1.chain)
    ref = oracle_sample(oracle, w["target"], Z1[-51:], N, 10, G, None, w["eps_scale"], 2.38, 2,
                        X0=mc1.chain[:, :, -1], lp0=mc1.log_objcurrent, rng_offset=G)
    assert np.array_equal(mc2.chain[:, :, G:], ref["chain"])
    # same seed, resumed: the chains' streams continue, so (G then G more) == 2G in one go when K | G
    a, Za = demc.demcz_sample(w["target"], w["Zinit"][:50], N, 10, 2 * G, 1, [range(d)], w["eps_scale"], 2.38, seed=1, **kw)
    b, Zb = demc.demcz_sample(w["target"], Z1, N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, seed=1, prevrun=mc1, **kw)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(Za, Zb)


def test_checkpoint_resume_continues_the_random_streams(oracle, tmp_path):
    """A checkpoint keeps one generation of history; the prevrun it restores must still resume the chains'
    Philox streams where they stopped -- WITHOUT the caller passing rng_offset (it used to replay
    generations 2..G+1 of the first run's draws silently).  Twice in a row: G + G + G == 3G in one go."""
    d, N, G = 5, 6, 30
    w = demc.workloads.mvnormal_problem(d, N)
    kw = dict(verbose=False, engine_factory=OracleEngine)
    args = (N, 10, G, 1, [range(d)], w["eps_scale"], 2.38)
    one, Zone = demc.demcz_sample(w["target"], w["Zinit"][:50], N, 10, 3 * G, 1, [range(d)], w["eps_scale"], 2.38, seed=4, **kw)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"][:50], *args, seed=4, **kw)
    assert mc.generations_drawn == G
    for leg in (1, 2):
        demc.save_checkpoint(tmp_path / f"ck{leg}.npz", mc, Z, seed=4)
        prev, Zc, done, seed = demc.load_checkpoint(tmp_path / f"ck{leg}.npz")
        assert done == leg * G and prev.chain.shape[2] == 1 and prev.generations_drawn == leg * G and seed == 4
        mc, Z = demc.demcz_sample(w["target"], Zc, *args, prevrun=prev, seed=seed, **kw)
        assert mc.generations_drawn == (leg + 1) * G
        # demcz.jl:58-59 concatenates prevrun's (one-generation) history in front of the new one
        assert np.array_equal(mc.chain[:, :, 1:], one.chain[:, :, leg * G:(leg + 1) * G])
    assert np.array_equal(Z, Zone) and np.array_equal(mc.Xcurrent, one.Xcurrent)
    # an explicit rng_offset still wins (e.g. to decorrelate on purpose)
    other, _ = demc.demcz_sample(w["target"], Zc, *args, prevrun=prev, seed=seed, rng_offset=10 ** 6, **kw)
    assert not np.array_equal(other.chain[:, :, 1:], mc.chain[:, :, 1:])


def test_autostop_truncates_at_first_passing_check(oracle):
    """demcz.jl:39-53 through the opts path (the canonical defaults, Q8)."""
    d, N = 5, 32
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=3000, eps_scale=w["eps_scale"], verbose=False,
                        autostop="Rhat", autostop_every=500, autostop_Rhat=1.2)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=4, engine_factory=OracleEngine)
    G = mc.chain.shape[2]
    assert G % 500 == 0 and G < 3000
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 4)
    assert np.array_equal(mc.chain, ref["chain"])
    assert Z.shape[0] == ref["M"]                                    # Z[1:M,:], demcz.jl:51
    for g in range(500, G + 1, 500):
        r = oracle.rhat_gelman(ref["chain"][:, :, g - 500:g])
        assert (np.max(r) < 1.2) == (g == G)
    opts.autostop = "no"
    mc3, _ = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=4, engine_factory=OracleEngine)
    assert mc3.chain.shape[2] == 3000


def test_anneal_schedule_and_gamma_adaptation(oracle):
    """demcz_anneal.jl:39-57: T(ig) = T0 (TN/T0)^(ig/Ng), gamma halves when the windowed
    accept ratio < 0.1, x1.5 when > 0.5 (bounded by min/max gamma)."""
    d, N, G, ae = 10, 16, 300, 100
    w = demc.workloads.iso_quad_problem(d, N)
    mc, Z = demc.demcz_anneal(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False,
                              T0=2, TN=1e-4, seed=6, adaptγ={"adapt_every": ae}, engine_factory=OracleEngine)
    # replay with the oracle, adapting gamma by the reference's formula
    M0 = w["Zinit"].shape[0]
    Mcap = M0 + -(-N * G // 10)
    prob = oracle.Problem(N, d, 10, Mcap, w["eps_scale"], 6, target=w["target"].spec())
    X = np.array(w["Zinit"][M0 - N:], order="F")
    lp = oracle.logp(prob, X)
    Zo = np.zeros((Mcap, d), order="F")
    Zo[:M0] = w["Zinit"]
    M, gam, chains, lobjs, gammas = M0, 2.38, [], [], []
    for s in range(0, G, ae):
        temps = np.array([2 * (1e-4 / 2) ** (g / G) for g in range(s + 1, s + ae + 1)])
        M, ch, lo, _ = oracle.run(prob, X, lp, Zo, M, s + 1, s + ae, gam, temperature=temps)
        chains.append(ch); lobjs.append(lo); gammas.append(gam)
        ratio = (np.diff(lo, axis=1) != 0).sum() / (N * ae)          # demcz_anneal.jl:50
        if ratio < 0.1:
            gam = max(0.1, gam * 0.5)
        elif ratio > 0.5:
            gam = min(4.0, gam * 1.5)
    assert np.array_equal(mc.chain, np.concatenate(chains, axis=2))
    assert np.array_equal(mc.log_obj, np.concatenate(lobjs, axis=1))
    assert len(set(gammas)) > 1, "test should exercise an adaptation step"
    assert mc.log_obj.max() > -0.2                                   # what test/test_anneal.jl:31 meant to assert


def test_anneal_compat_serial_temperature(oracle):
    """Q9: the serial reference ignores T0/TN/Ngeneration: T(ig) = tempbaseline(ig, 1000, 1, 1e-3)."""
    d, N, G = 10, 8, 30
    w = demc.workloads.iso_quad_problem(d, N)
    mc, _ = demc.demcz_anneal(w["target"], w["Zinit"], N, 10, G, 1, [range(d)], w["eps_scale"], 2.38, verbose=False,
                              T0=5, TN=0.0, seed=6, adaptγ={"adapt": False}, compat_serial_temp=True,
                              engine_factory=OracleEngine)
    temps = np.array([1.0 * (1e-3 / 1.0) ** (g / 1000) for g in range(1, G + 1)])
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None, w["eps_scale"], 2.38, 6, temperature=temps)
    assert np.array_equal(mc.chain, ref["chain"])


@pytest.mark.parametrize("shards", [2, 4])
def test_in_process_shards_are_invariant(oracle, shards):
    """Chains sharded S ways (host-driven exchange): identical bits to the unsharded run, because
    chain c's stream is Philox subsequence c whatever shard holds it."""
    d, N, G = 5, 16, 45
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=20, autostop_Rhat=1.0)     # checks run, never stop
    a, Za = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=8, engine_factory=OracleEngine)
    sh = demc.Sharding(rank=0, world_size=1, mode="host", local_shards=shards)
    b, Zb, runner = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=8, engine_factory=OracleEngine, sharding=sh,
                                      return_runner=True)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(Za, Zb)
    assert np.allclose(runner.rhat(1, 40), oracle.rhat_gelman(a.chain[:, :, :40]), rtol=1e-12)
    assert np.array_equal(runner.changed(1, G), oracle_sample(oracle, w["target"], w["Zinit"], N, 10, G, None,
                                                            w["eps_scale"], 2.38, 8)["changed"])
    runner.close()


def test_blocks_to_csr(demc):
    from demc_jl_amd.engine import blocks_to_csr
    offs, idx = blocks_to_csr([range(0, 2), [4, 2], [3]], 5)
    assert list(offs) == [0, 2, 4, 5] and list(idx) == [0, 1, 4, 2, 3]
    with pytest.raises(ValueError):
        blocks_to_csr([[0, 5]], 5)
