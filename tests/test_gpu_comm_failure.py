"""GPU: what a sharded handle does when the exchange makes no progress (a dead or stalled peer rank).  The reference's
pmap (src/demcz.jl:137) throws when a worker dies; here every host-side wait behind an RCCL collective has a deadline:
past it both communicators are aborted (ncclCommAbort) and DEMCZ_ERR_COMM comes back -- never a silent hang.  On a
one-GPU box the stalled peer is played by demcz_debug_stall_exchange (a kernel that holds the collective's stream back)."""
import time

import numpy as np
import pytest

from helpers import oracle_sample

pytestmark = pytest.mark.gpu


def _sharded_engine(demc, lag, G=400, N=256, d=5, K=10, seed=5):
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=seed,
                       target=w["target"])
    e.comm_init(e.comm_unique_id(), 1, 0)          # a communicator of one rank: every RCCL call of the data path runs
    if lag:
        e.set_append_lag(lag)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    return e, w


@pytest.mark.parametrize("lag", [0, 2])
@pytest.mark.parametrize("entry", ["run+synchronize", "run_checked"])
def test_stalled_exchange_hits_the_deadline_and_aborts(demc, lag, entry):
    from demc_jl_amd._lib import ERR_COMM, DemczError
    e, w = _sharded_engine(demc, lag)
    e.run(1, 40, w["gamma"])
    e.synchronize()                                # healthy so far
    e.set_comm_timeout(50)
    e.debug_stall_exchange(4000)                   # the next collective is held back for 4 s: far beyond the deadline
    t0 = time.perf_counter()
    with pytest.raises(DemczError) as ei:
        if entry == "run_checked":
            e.run_checked(41, 400, w["gamma"], 40, 0.0)
        else:
            e.run(41, 400, w["gamma"])
            e.synchronize()
    dt = time.perf_counter() - t0
    assert ei.value.code == ERR_COMM and "aborted" in str(ei.value), str(ei.value)
    assert dt < 3.0, f"the failure took {dt:.2f} s to surface (deadline 50 ms, bounded drain 2 s)"
    for call in (lambda: e.run(401, 402, w["gamma"]), e.synchronize, lambda: e.get_history(1, 10), e.get_state):
        with pytest.raises(DemczError) as ej:     # the handle is dead: every call says so at once
            call()
        assert ej.value.code == ERR_COMM
    t1 = time.perf_counter()
    e.close()                                      # and destroying it does not hang either
    assert time.perf_counter() - t1 < 3.0


@pytest.mark.parametrize("lag", [0, 2])
def test_slow_exchange_within_the_deadline_is_just_slow(demc, oracle, lag):
    """A collective held back for 30 ms under a 5 s deadline: the polling waits change nothing in the results."""
    from oracle_engine import OracleEngine
    G, N, d, K, seed = 200, 256, 5, 10, 5
    e, w = _sharded_engine(demc, lag, G=G, N=N, seed=seed)
    e.set_comm_timeout(5000)
    e.debug_stall_exchange(30)
    e.run_checked(1, G, w["gamma"], 50, 0.0)
    ch, lo = e.get_history(1, G)
    X, lp, Z, M = e.get_state()
    e.close()
    if lag == 0:
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
        assert np.array_equal(ch, ref["chain"]) and np.array_equal(Z, ref["Z"])
    else:
        opts = demc.demcopt(d, N=N, K=K, Ngeneration=G, eps_scale=w["eps_scale"], verbose=False, autostop="no")
        sh = demc.Sharding(mode="host", local_shards=1, host_exchange_always=True)
        mc, Zr = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=seed, sharding=sh, append_lag=lag, engine_factory=OracleEngine)
        assert np.array_equal(ch, mc.chain) and np.array_equal(Z, Zr)
