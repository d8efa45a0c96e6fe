"""GPU: replicated archives whose K-boundary rows are handed over INSIDE the launches (demcz_kernels_rec.h, live_publish) -- the
schedule of a multi-GPU run without its collective: every shard's publisher waves store a boundary's rows into every replica,
readers poll their own.  The reference's counterpart is the archive its workers share under pmap (src/demcz.jl:88-91, 137).

A one-GPU box cannot put the replicas on different GPUs, so they are R handles of this process on the one device
(demcz_peer_group), each with its own replica and 1/R of the chains, LIVE-co-resident: the same kernels, the same addressing
(row = M + boundary * N_total + rank * N + chain), the same waits -- only the stores to "peers" do not cross xGMI.  Every case
is compared with the ORACLE's unsharded run, bit for bit: results must not depend on the sharding.  The IPC half of the
multi-GPU set-up (fine-grained archive, hipIpcGetMemHandle, the all-gather of the handles) runs with a one-rank communicator."""
import os

import numpy as np
import pytest

from helpers import SPLIT, SPLIT_WAVE, oracle_sample

pytestmark = pytest.mark.gpu

THREADS = max(1, min(len(os.sched_getaffinity(0)), 8))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
BLOCKS_D20 = [list(range(0, 5)), list(range(5, 10)), list(range(10, 15)), list(range(15, 20))]


def _communicate_all(procs, timeout):
    """communicate() with every child; whatever happens (a rank that hangs, an exception here) no child outlives the test: a
    rank left polling on the GPU would hold it for the rest of the session."""
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
    return outs


def _group(demc, w, R, n_loc, d, K, G, seed, blocks=None, lanes=0, group=True):
    N = R * n_loc
    M0 = w["Zinit"].shape[0]
    X0 = w["Zinit"][-N:]
    es = []
    for r in range(R):
        e = demc.HipEngine(N=n_loc, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=blocks or [range(d)],
                           eps_scale=w["eps_scale"], seed=seed, target=w["target"], chain_id0=r * n_loc, lanes_per_chain=lanes)
        e.set_state(X0[r * n_loc:(r + 1) * n_loc], None, w["Zinit"])
        es.append(e)
    if group:
        demc.HipEngine.peer_group(es)
    return es


def _run(es, pieces, gamma, temperature=None):
    g = 1
    for n in pieces:
        for e in es:                              # every member is given the call before any member is asked for anything
            e.run(g, g + n - 1, gamma, None if temperature is None else temperature[g - 1:g + n - 1])
        g += n
    return g - 1


def _collect(es, G):
    ch = np.concatenate([e.get_history(1, G)[0] for e in es], axis=0)
    lo = np.concatenate([e.get_history(1, G)[1] for e in es], axis=0)
    sts = [e.get_state() for e in es]
    X = np.concatenate([s[0] for s in sts], axis=0)
    lp = np.concatenate([s[1] for s in sts])
    return ch, lo, X, lp, [s[2] for s in sts], [s[3] for s in sts]


def _check(es, G, ref):
    ch, lo, X, lp, Zs, Ms = _collect(es, G)
    assert np.array_equal(ch, ref["chain"]), "chain history differs from the oracle's unsharded run"
    assert np.array_equal(lo, ref["log_obj"])
    assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"])
    for Z, M in zip(Zs, Ms):                      # every replica is the whole archive
        assert M == ref["M"] and np.array_equal(Z, ref["Z"]), "a replica differs from the oracle's archive"


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("d,G,pieces", [(5, 1000, [1000]), (5, 615, [200, 5, 410]), (20, 600, [250, 350])])
def test_replica_group_equals_oracle(demc, oracle, R, d, G, pieces):
    """C2's shape (d = 5: window_kernel_ps2, and window_kernel_ps for the irregular pieces) and C4's shard shape (d = 20:
    window_kernel_pw): 1024 chains as R replicas of 1024 / R, K = 10; every launch LIVE, a handful of launches per run."""
    N, K, seed = 1024, 10, 900 + d + R
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, R, N // R, d, K, G, seed)
    assert all(e.info()["lanes_per_chain"] == SPLIT_WAVE for e in es)
    assert all(e.peer_status() == (1, R - 1) for e in es)
    _run(es, pieces, w["gamma"])
    for e in es:
        e.synchronize()
    assert all(e.live_status() == (True, 0) for e in es), [e.live_status() for e in es]
    assert all(e.info()["window_launches"] <= 2 * len(pieces) + 2 for e in es), [e.info() for e in es]
    if d == 5 and pieces == [1000]:
        assert all(e.kernel_counts()["ps2"] >= 1 for e in es)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _check(es, G, ref)
    for e in es:
        e.close()


@pytest.mark.parametrize("R,d", [(2, 5), (4, 5), (2, 20)])
def test_replica_group_forced_timeout_is_redone_in_lockstep(demc, oracle, R, d):
    """A poll limit of 1 on ONE member makes its first wait for a row give up.  The whole group then goes back to the state before
    the unverified calls and redoes them in lockstep, one launch per K-window, rows appended from the members' state buffers;
    calls made afterwards are executed the same way at the next verification.  Same bits as the oracle throughout."""
    N, K, G1, G2, seed = 512, 2, 120, 60, 77 + R
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, R, N // R, d, K, G1 + G2, seed)
    es[R - 1].set_live_spin_limit(1)
    es[0].set_live_rearms(0)                         # (of the whole group; re-arming: test_replica_group_goes_live_again)
    _run(es, [50, 70], w["gamma"])
    es[0].synchronize()                              # verifies the GROUP: rollback and lockstep redo of both calls
    assert all(e.live_status()[1] == 1 and not e.live_status()[0] for e in es), [e.live_status() for e in es]
    ref1 = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G1, None, w["eps_scale"], w["gamma"], seed)
    _check(es, G1, ref1)
    for e in es:                                     # deferred from now on: logged, executed at the next verification
        e.run(G1 + 1, G1 + G2, w["gamma"])
    ref2 = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G1 + G2, None, w["eps_scale"], w["gamma"], seed)
    _check(es, G1 + G2, ref2)
    for e in es:
        e.close()


def test_replica_group_member_that_finds_the_launch_failed(demc, oracle):
    """Fault injection -1 (tests/test_gpu_live.py) on one member of a group: its launch finds the error word set and every wave
    of it leaves at once -- the other members' waits for ITS rows then time out (poll limit lowered so that they do so quickly),
    and the group redo must start from every member's exact state."""
    R, N, d, K, G, seed = 2, 512, 5, 10, 200, 31
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, R, N // R, d, K, G, seed)
    es[1].debug_set_live_fault(-1, 1)
    es[0].set_live_spin_limit(64)
    _run(es, [G], w["gamma"])
    es[1].synchronize()
    assert all(e.live_status()[1] == 1 for e in es)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    _check(es, G, ref)
    for e in es:
        e.close()


def test_replica_group_block_updates_and_regression(demc, oracle):
    """The lane-cooperative consumers publish through live_publish too: block updates at d = 20 (window_kernel_mlb) as 2 x 512
    chains, and the tempered regression kernel (window_kernel_lr8s / lr16) as 2 x 256."""
    K, seed = 10, 5
    w = demc.workloads.mvnormal_problem(20, 1024)
    es = _group(demc, w, 2, 512, 20, K, 300, seed, blocks=BLOCKS_D20)
    _run(es, [300], w["gamma"])
    for e in es:
        e.synchronize()
    assert all(e.live_status() == (True, 0) for e in es)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], 1024, K, 300, BLOCKS_D20, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _check(es, 300, ref)
    for e in es:
        e.close()
    w = demc.workloads.linreg_problem(10, 512)
    G = 200
    temps = np.array([demc.tempbaseline(g, G, 3.0, 1e-3) for g in range(1, G + 1)])
    es = _group(demc, w, 2, 256, 10, K, G, seed)
    _run(es, [G], 0.5, temperature=temps)
    for e in es:
        e.synchronize()
    assert all(e.live_status() == (True, 0) for e in es)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], 512, K, G, None, w["eps_scale"], 0.5, seed, temperature=temps, threads=THREADS)
    _check(es, G, ref)
    for e in es:
        e.close()


def test_replica_group_through_the_sampler_surface(demc, oracle):
    """demcz_sample(..., sharding=Sharding(mode="peer", local_shards=4)): the reference surface over a replica group, R-hat
    autostop included (partial moments summed over the members on the host)."""
    d, N, K, G, seed = 5, 512, 10, 400, 12
    w = demc.workloads.mvnormal_problem(d, N)
    sh = demc.Sharding(mode="peer", local_shards=4)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], N, K, G, 1, [range(d)], w["eps_scale"], w["gamma"], verbose=False,
                              seed=seed, sharding=sh, autostop="Rhat", autostop_Rhat=0.5, autostop_every=100)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(mc.chain, ref["chain"]) and np.array_equal(mc.log_obj, ref["log_obj"]) and np.array_equal(Z, ref["Z"])


def test_replica_group_rules(demc):
    """What demcz_peer_group refuses, and what is left of a group when a member is destroyed."""
    d, N, K, G = 5, 256, 10, 40
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, 2, 128, d, K, G, 3, group=False)
    with pytest.raises(demc.DemczError):
        demc.HipEngine.peer_group([es[0], es[0]])
    with pytest.raises(demc.DemczError):
        demc.HipEngine.peer_group([es[1], es[0]])            # chain_id0 must be r * N in rank order
    demc.HipEngine.peer_group(es)
    with pytest.raises(demc.DemczError):
        demc.HipEngine.peer_group(es)                        # already grouped
    with pytest.raises(demc.DemczError):
        es[0].run_checked(1, G, w["gamma"], 20)              # members are driven call by call
    es[0].run(1, G, w["gamma"])
    with pytest.raises(demc.DemczError):
        es[0].get_history(1, G)                              # member 1 has not been given the call
    es[1].run(1, G, w["gamma"])
    ch0, _ = es[0].get_history(1, G)
    es[0].close()                                            # ends the group
    ch1, _ = es[1].get_history(1, G)                         # what was verified stays readable
    assert np.isfinite(ch0).all() and np.isfinite(ch1).all()
    with pytest.raises(demc.DemczError):
        es[1].run(G + 1, G + 10, w["gamma"])
    es[1].close()


def test_ipc_set_up_with_a_one_rank_communicator(demc, oracle, monkeypatch):
    """demcz_comm_init's second half on a communicator of one rank (DEMCZ_PEER_SELF): the archive moves into a fine-grained
    allocation with its contents (set_state ran before), is exported over IPC, the handles travel through ncclAllGather, the
    agreement through ncclAllReduce; the sharded handle then keeps its LIVE launches (no all-gather per K-window) and the error
    words are max-reduced over the communicator at every verification.  Results: the oracle's."""
    monkeypatch.setenv("DEMCZ_PEER_SELF", "1")
    for d, G in ((5, 1000), (20, 300)):
        N, K, seed = 512, 10, 60 + d
        w = demc.workloads.mvnormal_problem(d, N)
        M0 = w["Zinit"].shape[0]
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                           seed=seed, target=w["target"])
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        e.comm_init(e.comm_unique_id(), 1, 0)
        assert e.peer_status() == (2, 0)
        e.run(1, G // 2, w["gamma"])
        _, trace, _ = e.run_checked(G // 2 + 1, G, w["gamma"], 100)
        assert e.live_status() == (True, 0) and e.info()["window_launches"] <= 6
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        e.close()
        ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
        assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"]) and np.array_equal(Z, ref["Z"]) and M == ref["M"]
        assert np.array_equal(X, ref["X"]) and np.array_equal(lp, ref["logp"])
    # a forced time-out in that mode: every rank rolls back, the redo exchanges through ncclAllGather
    d, N, K, G, seed = 5, 256, 2, 80, 9
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                       seed=seed, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.comm_init(e.comm_unique_id(), 1, 0)
    e.set_live_spin_limit(1)
    e.run(1, G, w["gamma"])
    e.synchronize()
    assert e.live_status() == (False, 1)
    ch, _ = e.get_history(1, G)
    _, _, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(Z, ref["Z"])


@pytest.mark.parametrize("d,pieces", [(5, "600,400"), (20, "300")])
def test_two_processes_on_one_gpu_hand_rows_over_through_ipc(demc, oracle, tmp_path, d, pieces):
    """The cross-PROCESS half of the multi-GPU path, as far as one GPU can show it: two processes, each with its own HIP context,
    its own archive replica in fine-grained memory and half of the chains; each opens the other's archive with
    hipIpcOpenMemHandle (demcz_peer_export / demcz_peer_attach, the 64-byte handles carried by gloo) and its publisher waves
    store every boundary's rows into both replicas from inside its launches; readers poll their own replica.  What this cannot
    show is the store crossing xGMI to ANOTHER GPU's memory.  Both ranks' results: the oracle's unsharded run, bit for bit."""
    import socket
    import subprocess
    import sys
    from pathlib import Path
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    case = str(Path(__file__).resolve().parent / "peer_ipc_case.py")
    procs = [subprocess.Popen([sys.executable, case, str(r), str(world), str(port), str(tmp_path), str(d), pieces],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = _communicate_all(procs, 300)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (p.returncode, so[-1500:], se[-3000:])
    N, K, seed = 1024, 10, 2024 + d
    G = sum(int(v) for v in pieces.split(","))
    w = demc.workloads.mvnormal_problem(d, N)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in rs:
        assert (int(r["mode"]), int(r["peers"])) == (3, 1) and int(r["live"]) == 1 and int(r["redos"]) == 0
        assert int(r["launches"]) <= 6, "LIVE launches through the boundaries: a handful per run"
        assert int(r["M"]) == ref["M"] and np.array_equal(r["Z"], ref["Z"]), "a replica differs from the oracle's archive"
    assert np.array_equal(np.concatenate([r["chain"] for r in rs], axis=0), ref["chain"])
    assert np.array_equal(np.concatenate([r["log_obj"] for r in rs], axis=0), ref["log_obj"])
    assert np.array_equal(np.concatenate([r["X"] for r in rs], axis=0), ref["X"])


def test_replica_group_goes_live_again(demc, oracle):
    """A forced time-out on one member (fault injection, fires once): the group rolls back, executes what was logged in lockstep --
    and from the next call on its members publish into each other's replicas from inside their launches again (round 5)."""
    R, N, d, K, G1, G2, seed = 2, 512, 5, 10, 300, 300, 5
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, R, N // R, d, K, G1 + G2, seed)
    es[1].debug_set_live_fault(1, 1)
    _run(es, [G1], w["gamma"])
    es[0].synchronize()
    assert all(e.live_status()[1] == 1 for e in es) and all(e.live_rearms() == (1, 2) for e in es), [e.live_rearms() for e in es]
    n0 = [e.info()["window_launches"] for e in es]
    for e in es:
        e.run(G1 + 1, G1 + G2, w["gamma"])
    for e in es:
        e.synchronize()
    assert all(e.live_status() == (True, 1) for e in es), [e.live_status() for e in es]
    assert all(e.info()["window_launches"] - n <= 4 for e, n in zip(es, n0)), "LIVE launches again: a handful per call"
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G1 + G2, None, w["eps_scale"], w["gamma"], seed)
    _check(es, G1 + G2, ref)
    for e in es:
        e.close()


def test_rearm_with_the_ipc_set_up_of_a_one_rank_communicator(demc, oracle, monkeypatch):
    """The sharded handle's path (peer mode 2): error words max-reduced, the failed row min-reduced, the redo through
    ncclAllGather up to the failed slab, the ranks' agreement to go LIVE again (a min-reduction), the stream-ordered meeting
    before the first publishing launch -- on a communicator of one rank.  6-slab demcz_run_checked, fault in slab 3."""
    monkeypatch.setenv("DEMCZ_PEER_SELF", "1")
    N, d, K, every, seed = 512, 5, 10, 200, 29
    G = 6 * every
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                       seed=seed, target=w["target"])
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.comm_init(e.comm_unique_id(), 1, 0)
    assert e.peer_status() == (2, 0) and e.peer_ping()[0] == 1
    e.debug_set_live_fault(1, 2 * every + 1)
    g_stop, trace, _ = e.run_checked(1, G, w["gamma"], every, 0.0)
    assert e.live_status() == (True, 1) and e.live_rearms() == (1, 2)
    launches = e.info()["window_launches"]
    assert 3 * every // K + 3 <= launches <= 3 * every // K + 14, launches
    ch, lo = e.get_history(1, G)
    _, _, Z, M = e.get_state()
    e.close()
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(ch, ref["chain"]) and np.array_equal(lo, ref["log_obj"]) and np.array_equal(Z, ref["Z"]) and M == ref["M"]


def test_a_two_chain_handle_that_becomes_a_peer_runs_one_chain_per_wave(demc, oracle):
    """1024 < N <= 2048 chains per shard at d <= 5 selects two chains to a wave (window_kernel_ps2d), whose
    irregular launches cannot hand rows over; as a peer such a handle must not mix the two.  Two members of 1100 chains, calls
    that are not multiples of five: every replica must hold the oracle's archive (no sentinel row read as data, none missing)."""
    R, n_loc, d, K, seed = 2, 1100, 5, 10, 3
    N = R * n_loc
    pieces = [103, 7, 190]
    G = sum(pieces)
    w = demc.workloads.mvnormal_problem(d, N)
    es = _group(demc, w, R, n_loc, d, K, G, seed)
    _run(es, pieces, w["gamma"])
    for e in es:
        e.synchronize()
    assert all("ps2d" not in e.kernel_name() for e in es), [e.kernel_name() for e in es]
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    _check(es, G, ref)
    for e in es:
        e.close()


@pytest.mark.parametrize("d,G", [(5, 1000), (20, 400)])
def test_eight_replicas_concurrently_on_one_gpu(demc, oracle, tmp_path, d, G):
    """C4's fan-out -- eight replicas, seven peer stores per published element, eight launches polling each other -- as far as one
    GPU can show it: a replica group of R = 8 x 128 chains.  HIP multiplexes a process's streams over four hardware queues by
    default, on which eight launches that wait for each other cannot all run; GPU_MAX_HW_QUEUES = 8, set before the process's
    first GPU call, lifts that -- hence a fresh child process (tests/peer_group_r8_case.py).  Bit-equal to the oracle's unsharded
    run, LIVE launches on every member (no lockstep fall-back)."""
    import subprocess
    import sys
    from pathlib import Path
    case = str(Path(__file__).resolve().parent / "peer_group_r8_case.py")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    p = subprocess.Popen([sys.executable, case, str(tmp_path), str(d), str(G)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    (so, se), = _communicate_all([p], 300)
    assert p.returncode == 0, (p.returncode, so[-1500:], se[-3000:])
    r = np.load(tmp_path / "r8.npz")
    N, K, seed = 1024, 10, 808 + d
    w = demc.workloads.mvnormal_problem(d, N)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, K, G, None, w["eps_scale"], w["gamma"], seed, threads=THREADS)
    assert list(r["live"]) == [1] * 8 and list(r["redos"]) == [0] * 8, (r["live"], r["redos"])
    assert max(r["launches"]) <= 6, "LIVE launches through the boundaries on every member"
    assert np.array_equal(r["chain"], ref["chain"]) and np.array_equal(r["log_obj"], ref["log_obj"])
    assert np.array_equal(r["X"], ref["X"]) and all(int(m) == ref["M"] for m in r["M"])
    assert np.array_equal(r["Z0"], ref["Z"]) and np.array_equal(r["Z7"], ref["Z"])
