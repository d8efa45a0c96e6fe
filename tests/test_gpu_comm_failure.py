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
def test_stalled_exchange_hits_the_deadline_and_aborts(lag, entry):
    """Each scenario in a fresh process (tests/comm_failure_case.py): what ncclCommAbort leaves behind is not a state to run the
    next scenario in."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from demc_jl_amd._lib import ERR_COMM
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "comm_failure_case.py"), str(lag), entry],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["raised"] == ERR_COMM and "aborted" in out["message"], out
    assert out["seconds_to_surface"] < 2.5, out         # deadline 50 ms + the bounded drain (1 s) of the aborted streams
    assert [c for _, c in out["later"]] == [ERR_COMM] * 4, out
    assert out["seconds_to_close"] < 3.0, out


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
