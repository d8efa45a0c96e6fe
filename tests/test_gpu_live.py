"""GPU: the in-launch row hand-off (LIVE launches of the split layout) and what happens when it fails.

A LIVE launch runs through many K boundaries; its waves hand the appended rows to each other through the
archive itself (demcz_kernels_rec.h).  A wave that does not see a row within its poll limit gives up and
flags the launch; the library then redoes everything since the last verified point with one launch per
K-window and keeps the handle in that mode.  These tests lower the poll limit to 1 (diagnostic entry point
demcz_set_live_spin_limit), so that the very first wait anywhere times out, and require the results to be
the oracle's, bit for bit, all the same."""
import numpy as np
import pytest

from helpers import SPLIT, SPLIT_WAVE, oracle_sample

pytestmark = pytest.mark.gpu


def _engine(demc, w, N, d, K, G, seed, lanes=0):
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)],
                       eps_scale=w["eps_scale"], seed=seed, target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    return e


@pytest.mark.parametrize("layout,d", [(SPLIT, 5), (SPLIT_WAVE, 5), (SPLIT, 20), (SPLIT_WAVE, 20)])
@pytest.mark.parametrize("K", [1, 3])
def test_forced_handoff_timeout_is_redone_bit_exact(demc, oracle, K, layout, d):
    """demcz_run + a synchronising call: poll limit 1 makes a wave give up at its first wait (K = 1: every
    generation draws from rows appended one generation earlier, so there are waits at once).  Every LIVE consumer:
    eight replicated lanes / one wave per chain at d = 5, sixteen cooperating lanes / one wave per chain at d = 20."""
    N, G, seed = 512, 120, 41
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, G, seed, layout)
    assert e.info()["lanes_per_chain"] == layout
    e.set_live_spin_limit(1)
    e.set_live_rearms(0)                           # (this case: the fall-back for good; re-arming has its own tests below)
    e.run(1, 50, w["gamma"])
    e.run(51, G, w["gamma"])                       # two calls in the log: both are redone
    e.synchronize()                                # verifies, rolls back, redoes
    on, redos = e.live_status()
    assert redos == 1 and not on, "the poll limit of 1 must have forced the fall-back"
    chain, lobj = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    # the handle keeps working, one launch per K-window from now on
    e.set_live_spin_limit(0)
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(chain, ref["chain"]) and np.array_equal(lobj, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("layout", [SPLIT, SPLIT_WAVE])
def test_forced_handoff_timeout_inside_run_checked(demc, oracle, layout):
    """demcz_run_checked never returns statistics of a voided slab: the whole call is redone, and the trace,
    the stop decision and the state are those of a run that never used LIVE launches."""
    N, d, K, G, every, seed = 256, 5, 2, 400, 100, 43
    w = demc.workloads.mvnormal_problem(d, N)
    a = _engine(demc, w, N, d, K, G, seed, layout)
    a.set_live_spin_limit(1)
    a.set_live_rearms(0)
    ga, ta, la = a.run_checked(1, G, w["gamma"], every, 0.0)
    on, redos = a.live_status()
    assert redos == 1 and not on
    cha, loa = a.get_history(1, G)
    Xa, lpa, Za, Ma = a.get_state()
    a.close()
    b = _engine(demc, w, N, d, K, G, seed, layout)  # undisturbed twin
    gb, tb, lb = b.run_checked(1, G, w["gamma"], every, 0.0)
    onb, redosb = b.live_status()
    assert redosb == 0 and onb
    chb, lob = b.get_history(1, G)
    Xb, lpb, Zb, Mb = b.get_state()
    b.close()
    assert ga == gb and np.array_equal(ta, tb) and np.array_equal(la, lb)
    assert np.array_equal(cha, chb) and np.array_equal(loa, lob) and np.array_equal(Za, Zb) and Ma == Mb
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(cha, ref["chain"]) and np.array_equal(Za, ref["Z"])


def test_live_budget_per_device(demc, oracle):
    """The consumer waves of LIVE launches wait for each other, so what a process has in flight on a device must fit the chip at
    once: every handle claims its share of the device's LIVE capacity (round 4: a budget, where there used to be one owner).
    A 1024-chain handle takes all of it -- the next handle runs one launch per K-window -- two 256-chain handles share it; all
    of them bit-exact, and a share is free again once its handle is destroyed."""
    d, K, G = 5, 5, 60
    wa = demc.workloads.mvnormal_problem(d, 1024)
    w = demc.workloads.mvnormal_problem(d, 256)
    a = _engine(demc, wa, 1024, d, K, G, 7)
    b = _engine(demc, w, 256, d, K, G, 8)
    a.run(1, G, wa["gamma"])
    b.run(1, G, w["gamma"])
    a.synchronize(); b.synchronize()
    assert a.live_status()[0] and not b.live_status()[0]
    assert a.info()["window_launches"] < b.info()["window_launches"]
    cha, _ = a.get_history(1, G)
    chb, _ = b.get_history(1, G)
    a.close()
    c = _engine(demc, w, 256, d, K, 2 * G, 9)
    c.run(1, G, w["gamma"])
    b2 = _engine(demc, w, 256, d, K, G, 10)
    b2.run(1, G, w["gamma"])
    c.synchronize(); b2.synchronize()
    assert c.live_status()[0] and b2.live_status()[0], "two small handles share the budget"
    chc, _ = c.get_history(1, G)
    chb2, _ = b2.get_history(1, G)
    b.close(); c.close(); b2.close()
    for ch, seed, ww, n in ((cha, 7, wa, 1024), (chb, 8, w, 256), (chc, 9, w, 256), (chb2, 10, w, 256)):
        ref = oracle_sample(oracle, ww["target"], ww["Zinit"], n, K, G, None, ww["eps_scale"], ww["gamma"], seed)
        assert np.array_equal(ch, ref["chain"])


def test_discarded_speculative_slab_leaves_no_trace(demc):
    """demcz_run_checked with a threshold runs the next slab ahead of the decision and discards it on a stop:
    afterwards the history beyond g_stop reads as never written (zeros, demcz.jl:24), not as stale data."""
    N, d, K, every = 256, 5, 10, 200
    G = 5 * every
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, G, 3)
    g_stop, trace, last = e.run_checked(1, G, w["gamma"], every, 3.0)      # a generous threshold: stops early
    assert g_stop < G and trace[-1] < 3.0
    ch, lo = e.get_history(g_stop + 1, min(G, g_stop + every))
    assert not ch.any() and not lo.any()
    ch0, _ = e.get_history(1, g_stop)
    assert ch0.any()
    e.close()


@pytest.mark.parametrize("R,cnt,batched", [(2, 1, False), (8, 1, False), (2, 1, True), (2, 3, True), (8, 4, True), (3, 2, True)])
def test_sharded_scatter_kernels_for_R_ranks(demc, R, cnt, batched):
    """append_gathered_kernel / append_batch_kernel with R > 1 (otherwise reachable only behind ncclAllGather on R
    GPUs): a slab in the all-gather's layout [R][cnt][d][n_loc] must land in the archive in the order an unsharded run
    appends -- boundary, then rank, then chain (demcz.jl:88-91 for N = R * n_loc chains)."""
    n_loc, d = 37, 5
    w = demc.workloads.mvnormal_problem(d, n_loc)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=n_loc, d=d, K=10, Mcap=M0 + 2 * R * cnt * n_loc, Gcap=0, blockindex=[range(d)], eps_scale=w["eps_scale"],
                       seed=1, target=w["target"])
    e.set_state(w["Zinit"][-n_loc:], None, w["Zinit"])
    rng = np.random.default_rng(R * 100 + cnt)
    # states[s][r] = the n_loc x d block rank r holds at boundary s
    states = rng.standard_normal((cnt, R, n_loc, d))
    slab = np.empty((R, cnt, d, n_loc))
    for r in range(R):
        for s in range(cnt):
            slab[r, s] = states[s, r].T                       # each rank sends [cnt][d][n_loc] (column-major n_loc x d blocks)
    e.debug_append_slab(slab, R, cnt, batched)
    e.debug_append_slab(slab * 2.0, R, cnt, batched)            # a second exchange lands behind the first
    X, lp, Z, M = e.get_state()
    e.close()
    want = states.reshape(cnt * R * n_loc, d)                 # boundary-major, then rank, then chain
    assert M == M0 + 2 * R * cnt * n_loc
    assert np.array_equal(Z[:M0], w["Zinit"])
    assert np.array_equal(Z[M0:M0 + R * cnt * n_loc], want)
    assert np.array_equal(Z[M0 + R * cnt * n_loc:], 2.0 * want)


@pytest.mark.parametrize("kind,d,N,blocks,lanes", [
    ("mvn", 5, 100, None, 0), ("mvn", 5, 100, None, SPLIT), ("mvn", 3, 37, None, SPLIT_WAVE), ("mvn-T", 5, 100, None, SPLIT_WAVE),
    ("mvn-T", 4, 50, None, SPLIT), ("mvn-T", 20, 40, None, SPLIT_WAVE), ("mvn-T", 20, 40, None, SPLIT), ("mvn", 20, 40, None, SPLIT),
    ("mvn", 5, 100, None, 8), ("mvn", 5, 100, None, 1), ("mvn", 20, 40, None, 0), ("mvn", 20, 40, None, 16),
    ("mvn", 6, 50, [[0], [1, 2], [5, 3, 4]], 0), ("mvn", 6, 50, [[0], [1, 2], [5, 3, 4]], 8), ("mvn", 6, 50, [[0], [1, 2], [5, 3, 4]], 1),
    ("mvn", 7, 30, None, 1), ("linreg", 10, 70, None, 0), ("linreg", 10, 70, None, 16), ("iso", 10, 33, None, 0)])
def test_ballot_accept_counts_equal_history_counts(demc, oracle, kind, d, N, blocks, lanes):
    """The accept mask by wavefront ballot (north star): every window kernel counts "log_obj changed" per generation with one
    ballot + popcount into a scalar register; demcz_get_changed_total sums those words for ranges made of whole launches.
    Checker: the counts taken from the history (demcz_get_changed) and the oracle's."""
    G, K, seed = 90, 10, 5
    w = (demc.workloads.iso_quad_problem(d, N) if kind == "iso" else
         demc.workloads.linreg_problem(d, N, nobs=70) if kind == "linreg" else demc.workloads.mvnormal_problem(d, N))
    bl = blocks or [range(d)]
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=bl, eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"], lanes_per_chain=lanes)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    T = np.linspace(3.0, 0.01, G) if kind != "mvn" else None              # ("mvn-T": the tempered accept on the MvNormal target)
    cuts = [(1, 30), (31, 37), (38, 38), (39, 90)]                        # calls cut anywhere
    for a, b in cuts:
        e.run(a, b, w["gamma"], None if T is None else T[a - 1:b])
    hist = e.get_changed(1, G)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, [list(b) for b in bl] if blocks else None, w["eps_scale"],
                        w["gamma"], seed, temperature=T)
    assert np.array_equal(hist, ref["changed"])
    for a, b in [(1, 30), (1, 90), (31, 38), (2, 30), (32, 90), (39, 90), (40, 90)]:     # whole launches (-1st generation)
        tot, from_ballots = e.changed_total(a, b, with_source=True)
        assert tot == int(hist[a - 1:b].sum()), (a, b)
        assert from_ballots, (a, b)
    for a, b in [(5, 20), (1, 89), (3, 90)]:                              # not launch-aligned: counted from the history
        tot, from_ballots = e.changed_total(a, b, with_source=True)
        assert tot == int(hist[a - 1:b].sum()) and not from_ballots, (a, b)
    e.close()


def test_ballot_counts_without_a_history_window(demc, oracle):
    """Gcap = 0: no history on the device, the gamma adaptation's count still comes out of the kernels."""
    N, d, K, G = 64, 10, 10, 100
    w = demc.workloads.iso_quad_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=0, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=2,
                       target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    T = np.linspace(2.0, 0.1, G)
    e.run(1, 50, 2.38, T[:50])
    e.run(51, 100, 2.38, T[50:])
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], 2.38, 2, temperature=T)
    assert e.changed_total(52, 100, with_source=True) == (int(ref["changed"][51:].sum()), True)
    assert e.changed_total(1, 100) == int(ref["changed"].sum())
    e.close()


@pytest.mark.parametrize("layout,d", [(SPLIT_WAVE, 5), (SPLIT_WAVE, 20), (SPLIT, 5)])
@pytest.mark.parametrize("polls", [2, 6, 24])
def test_partial_timeouts_drain_and_are_redone(demc, oracle, layout, d, polls):
    """A poll limit of a few polls (not 1): short waits succeed, longer ones give up, so SOME waves abandon the launch while the
    others carry on -- through more boundaries, filling their publisher's LDS ring -- until they see the error word at a later
    wait or reach the launch's end.  The interleaving ADVICE r2 describes (a chain wave still going after the launch has been
    given up) must drain: the ring wait is bounded by the publisher wave, which never leaves before its chain waves
    (demcz_kernels_ps.h), and every row wait is bounded by its poll limit.  K = 2: a boundary every other generation."""
    N, K, G, seed = 1024, 2, 300, 47
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, G, seed, layout)
    assert e.info()["lanes_per_chain"] == layout
    e.set_live_spin_limit(polls)
    e.set_live_rearms(0)
    e.run(1, G, w["gamma"])
    e.synchronize()
    on, redos = e.live_status()
    assert redos in (0, 1) and on == (redos == 0)
    chain, lobj = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(chain, ref["chain"]) and np.array_equal(lobj, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("threshold", [0.0, 1.5])
def test_run_checked_of_more_than_256_slabs_redoes_from_its_entry(demc, oracle, threshold):
    """A demcz_run_checked call of 300 slabs (every = 4) whose hand-off fails in slab 280: the call must roll back to ITS entry
    and redo itself, not to a snapshot taken by a mid-call verification after 256 logged demcz_run calls.  With a
    threshold the speculative slab and the stop decision are part of what is redone."""
    N, d, K, every, seed = 256, 5, 10, 4, 53
    G = every * 300
    w = demc.workloads.mvnormal_problem(d, N)
    res = []
    for fault in (True, False):
        e = _engine(demc, w, N, d, K, G, seed, SPLIT_WAVE)
        if fault:
            e.debug_set_live_fault(1, every * 280)            # poll limit 1 from slab 280 on
            e.set_live_rearms(0)
        g_stop, trace, last = e.run_checked(1, G, w["gamma"], every, threshold)
        on, redos = e.live_status()
        assert (redos, on) == ((1, False) if fault else (0, True))
        ch, lo = e.get_history(1, g_stop)
        X, lp, Z, M = e.get_state()
        e.close()
        res.append((g_stop, trace, last, ch, lo, X, lp, Z, M))
    a, b = res
    assert a[0] == b[0] and a[8] == b[8]
    for x, y in zip(a[1:8], b[1:8]):
        assert np.array_equal(x, y, equal_nan=True)
    if threshold == 0.0:
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
        assert np.array_equal(a[3], ref["chain"]) and np.array_equal(a[7], ref["Z"])
    else:
        assert a[0] < every * 280 or a[0] == G      # (wherever it stopped, both runs agree; see above)


def test_regular_launches_take_the_steady_state_kernel(demc):
    """ps2_applicable (demcz_capi.hip): launches that start behind a K boundary with K and their length multiples of five run
    on window_kernel_ps2, any other launch on the general wave-per-chain kernel -- told apart by demcz_debug_kernel_counts."""
    N, d, K = 1024, 5, 10
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * 40, Gcap=300, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=5, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, 100, w["gamma"])
    e.run(101, 200, w["gamma"])
    c = e.kernel_counts()
    assert c["ps2"] == 2 and c["ps_general"] == 0 and c["other"] == 0, c
    e.run(201, 203, w["gamma"])          # three generations: not a multiple of five
    e.run(204, 300, w["gamma"])          # starts in the middle of a K-window
    c2 = e.kernel_counts()
    e.close()
    assert c2["ps2"] == 2 and c2["ps_general"] >= 2, c2


@pytest.mark.parametrize("K,first", [(5, 1), (10, 201)])
def test_wave_that_finds_the_launch_already_failed_still_writes_its_snapshot_row(demc, oracle, K, first):
    """window_kernel_ps2 writes the redo snapshot of the state itself (no copy launches in front of a slab).  A chain wave that
    becomes resident after another wave has already timed out leaves at once -- and must have written its chain's row of the
    snapshot before it does, or the rollback restores garbage.  Fault injection (demcz_debug_set_live_fault, polls = -1): the
    snapshot buffers are filled with NaN patterns, and the LIVE launch that starts at generation `first` finds the error word
    set, so EVERY wave takes the early exit.  The redo must then start from the exact state."""
    N, d, G, seed = 512, 5, 400, 47
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, G, seed, SPLIT_WAVE)
    if first > 1:
        e.run(1, first - 1, w["gamma"])
        e.synchronize()                            # verified: the snapshot of the next call is taken at generation `first`
    e.debug_set_live_fault(-1, first)
    e.set_live_rearms(0)
    e.run(first, G, w["gamma"])
    assert e.kernel_counts()["ps2"] >= 1, "the faulted launch must be the steady-state kernel's"
    e.synchronize()
    on, redos = e.live_status()
    assert redos == 1 and not on
    chain, lobj = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.isfinite(X).all() and np.isfinite(lp).all()
    assert np.array_equal(chain, ref["chain"]) and np.array_equal(lobj, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


# ---- re-arming (round 5): a time-out is not for ever ------------------------------------------------------------------------------
@pytest.mark.parametrize("layout,d", [(SPLIT_WAVE, 5), (SPLIT_WAVE, 20), (SPLIT, 5), (SPLIT, 20)])
def test_run_checked_goes_live_again_behind_the_failed_slab(demc, oracle, layout, d):
    """A hand-off that times out in slab 3 of a 6-slab demcz_run_checked (fault injection: poll limit 1 for launches that start at
    generation 2 * every + 1 or later; the fault switches itself off once it has fired).  The call rolls back to its entry and
    redoes itself: one launch per K-window through slab 3 -- the slab that holds the generation whose row never came -- and LIVE
    launches again from slab 4 on (demcz_capi.hip: live_rollback / live_try_rearm).  Told apart by the launch count: slabs 1-3 of
    the redo are `every / K` launches each, slabs 4-6 one each.  Results: those of an undisturbed twin and of the oracle."""
    N, K, every, seed = 512, 10, 200, 71
    G = 6 * every
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, G, seed, layout)
    e.debug_set_live_fault(1, 2 * every + 1)
    g_stop, trace, last = e.run_checked(1, G, w["gamma"], every, 0.0)
    on, redos = e.live_status()
    rearms, left = e.live_rearms()
    launches = e.info()["window_launches"]
    assert redos == 1 and rearms == 1 and left == 2 and on, (redos, rearms, left, on)
    # first attempt: at most 6 + a cold start; the redo: 3 * every / K one-window launches, then 3 (+ cold starts) LIVE ones
    assert 3 * every // K + 3 <= launches <= 3 * every // K + 6 + 8, launches
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    # ... and the handle keeps its LIVE launches afterwards
    before = e.info()["window_launches"]
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert g_stop == G and np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])
    b = _engine(demc, w, N, d, K, G, seed, layout)      # undisturbed twin: the same trace of statistics
    _, tb, lb = b.run_checked(1, G, w["gamma"], every, 0.0)
    b.close()
    assert np.array_equal(trace, tb) and np.array_equal(last, lb)


def test_rearm_after_plain_runs_and_its_bound(demc, oracle):
    """demcz_run calls + a synchronising call.  The poll limit of 1 stays (demcz_set_live_spin_limit), so every LIVE launch with a
    wait in it fails: the first verification redoes the calls up to the failed one per K-window and re-arms for the one behind it,
    which fails again, ... until the handle's re-arms (set to 2 here) are used up and it stays at one launch per K-window.  Every
    intermediate state is the oracle's; redos = 1 + re-arms."""
    N, d, K, seed = 512, 5, 2, 17
    w = demc.workloads.mvnormal_problem(d, N)
    G = 240
    e = _engine(demc, w, N, d, K, G, seed, SPLIT_WAVE)
    e.set_live_spin_limit(1)
    e.set_live_rearms(2)
    for a in range(1, G, 40):
        e.run(a, a + 39, w["gamma"])
    e.synchronize()
    on, redos = e.live_status()
    rearms, left = e.live_rearms()
    assert (redos, rearms, left, on) == (3, 2, 0, False), (redos, rearms, left, on)
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"]) and M == ref["M"] and np.array_equal(Z, ref["Z"])


def test_rearm_on_the_next_call_and_after_set_state(demc, oracle):
    """A failure in the LAST call made: nothing of the redo lies behind it, so the handle re-arms at the next call -- and a
    demcz_set_state in between (a new run, generations numbered from 1 again) does not lose the pending re-arm."""
    N, d, K, G, seed = 512, 5, 10, 400, 23
    w = demc.workloads.mvnormal_problem(d, N)
    e = _engine(demc, w, N, d, K, 2 * G, seed, SPLIT_WAVE)
    e.debug_set_live_fault(1, 1)
    e.run(1, G, w["gamma"])
    e.synchronize()
    assert e.live_status() == (False, 1) and e.live_rearms() == (0, 2)
    n0 = e.info()["window_launches"]
    e.run(G + 1, 2 * G, w["gamma"])
    e.synchronize()
    assert e.live_status() == (True, 1) and e.live_rearms() == (1, 2)
    assert e.info()["window_launches"] - n0 <= 4
    ch, _ = e.get_history(1, 2 * G)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, 2 * G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(ch, ref["chain"])
    # the same through set_state
    e.debug_set_live_fault(1, 1)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, G, w["gamma"])
    e.synchronize()
    assert e.live_status() == (False, 2) and e.live_rearms() == (1, 1)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    n0 = e.info()["window_launches"]
    e.run(1, G, w["gamma"])
    e.synchronize()
    assert e.live_status() == (True, 2) and e.live_rearms() == (2, 1) and e.info()["window_launches"] - n0 <= 4
    ch2, _ = e.get_history(1, G)
    e.close()
    assert np.array_equal(ch2, ref["chain"][:, :, :G])
