"""Host drivers with the reference's call surface.

Mirrors (same names, argument meaning, defaults and return shapes):

* ``DEMCopt`` / ``demcopt(Npar; ...)``          src/DEMC.jl:24-43
* ``MC``                                        src/DEMC.jl:10-15
* ``demcz_sample(logobj, Zmat, opts; prevrun)`` src/demcz.jl:1-3 and the positional driver :9-63
* ``demcz_anneal(logobj, Zmat, opts; ...)``     src/demcz_anneal.jl:14-65
* ``tempbaseline``                              src/demcz_anneal.jl:1-3

Everything from ``runchain!`` down runs on the GPU through ``include/demcz.h``; this file only
does what the reference's drivers do around that call: set-up, the slab loop, the autostop
test, gamma adaptation, truncation / prevrun concatenation of the results.

Differences from the reference that a caller can see:

* indices are 0-based (``blockindex=[range(0, Npar)]``), symbols are strings
  (``autostop="Rhat"`` / ``"no"``);
* chains start at the last N rows of ``Zmat`` as the reference documents (``init="last_rows"``);
  ``init="reference_zeros"`` reproduces what the serial driver actually does (Q1);
* all N chains of a generation see the same archive (Q2); the reference's serial driver lets
  chain ic+1 see chain ic's fresh row inside generations divisible by K;
* ``Z[:M]`` is returned (Q7); ``padded_Z=True`` returns the zero-padded matrix instead;
* randomness comes from ``seed`` (Philox4x32-10 streams per chain), not a global RNG.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from .engine import HipEngine
from .targets import is_device_target


# --------------------------------------------------------------------------------------------
# types (DEMC.jl:10-43)
# --------------------------------------------------------------------------------------------
@dataclass
class MC:
    """Result container, src/DEMC.jl:10-15.  Arrays are column-major like the Julia ones."""
    chain: np.ndarray            # N x d x G   parameter population for all generations
    log_obj: np.ndarray          # N x G       log obj along the chain
    Xcurrent: np.ndarray         # N x d       population
    log_objcurrent: np.ndarray   # N           log obj values
    # (not in the reference, which draws from a global RNG:) generations every chain's Philox stream has consumed when
    # this result was made.  None = the length of `chain` (right for any result that carries its whole history); a
    # checkpoint, which keeps only the last generation, records the true count here so that `prevrun=` resumes the
    # streams where they stopped instead of replaying them.
    rng_generations: Optional[int] = None
    # Philox blocks one generation of the run that made this result consumed per chain (it follows from its blocks: a block-step
    # of b parameters takes 1 + ceil(b / 2) + 1, or 3 for b = 1).  A `prevrun` resumed with OTHER blocks starts its streams at the
    # first generation boundary of the new size behind everything the previous run drew -- never inside it.  None: unknown (a
    # result made by hand, an old checkpoint): taken to be the new run's own.
    rng_blocks_per_generation: Optional[int] = None

    @property
    def generations_drawn(self):
        return int(self.chain.shape[2]) if self.rng_generations is None else int(self.rng_generations)


@dataclass
class DEMCopt:
    """Mutable options struct, src/DEMC.jl:24-39 (fields in the same order)."""
    N: int
    K: int
    Ngeneration: int
    Nblocks: int
    blockindex: list
    eps_scale: np.ndarray
    γ: float
    verbose: bool
    print_step: int
    T0: float
    TN: float
    autostop: str
    autostop_every: int
    autostop_Rhat: float

    @property
    def gamma(self):
        return self.γ

    @gamma.setter
    def gamma(self, v):
        self.γ = v


def demcopt(Npar, *, N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=None, eps_scale=None, γ=2.38,
            gamma=None, verbose=True, print_step=100, T0=3, TN=1e-3, autostop="Rhat", autostop_every=1000,
            autostop_Rhat=1.05) -> DEMCopt:
    """Keyword constructor with the defaults of src/DEMC.jl:41."""
    if blockindex is None:
        blockindex = [range(0, Npar)]
    if eps_scale is None:
        eps_scale = 1e-4 * np.ones(Npar)
    if gamma is not None:
        γ = gamma
    return DEMCopt(N, K, Ngeneration, Nblocks, list(blockindex), np.asarray(eps_scale, dtype=np.float64), float(γ),
                   verbose, print_step, float(T0), float(TN), _sym(autostop), autostop_every, autostop_Rhat)


def _sym(s):
    s = str(s).lstrip(":")
    if s not in ("Rhat", "no"):
        raise ValueError("autostop must be 'Rhat' or 'no'")
    return s


def tempbaseline(ig, Ng, T0, TN):
    """``T0*(TN/T0)^(ig/Ng)``, src/demcz_anneal.jl:1-3."""
    return T0 * (TN / T0) ** (ig / Ng)


DEFAULT_ADAPT = {"adapt": True, "minγ": 0.1, "maxγ": 4.0, "adapt_every": 500}   # demcz_anneal.jl:14


# --------------------------------------------------------------------------------------------
# sharding: one process per GPU, chains split in rank order, Z replicated
# --------------------------------------------------------------------------------------------
@dataclass
class Sharding:
    """How the N chains are spread.

    ``mode="rccl"``: one engine per process; the library all-gathers the K-boundary rows and
    all-reduces the R-hat moments itself (``demcz_comm_init``).  ``mode="host"``: the driver does
    the exchange through ``all_gather`` / ``all_reduce_sum`` callables on host arrays (any
    backend: torch.distributed gloo/nccl, MPI ...), optionally with several in-process shards
    (``local_shards`` > 1) -- used to prove that results do not depend on the sharding.  ``mode="peer"``: the
    ``local_shards`` engines of this process (one device) form a replica group (``demcz_peer_group``): every engine keeps its own
    archive replica and publishes its K-boundary rows into the others' from inside its launches -- the schedule a multi-GPU run
    uses over IPC (``demcz_comm_init``), rehearsed on one GPU; no exchange step on the host or in RCCL.
    """
    rank: int = 0
    world_size: int = 1
    mode: str = "host"
    local_shards: int = 1
    host_exchange_always: bool = False          # drive appends from the host even with one shard
    all_gather: Optional[Callable] = None       # np.ndarray -> list of np.ndarray, one per rank
    all_reduce_sum: Optional[Callable] = None   # np.ndarray -> np.ndarray
    broadcast_bytes: Optional[Callable] = None  # (bytes or None, src=0) -> bytes

    @property
    def total_shards(self):
        return self.world_size * self.local_shards


class _Runner:
    """The engines of this process plus the exchange between shards."""

    def __init__(self, engines, sharding: Optional[Sharding], K, N_total, d, append_lag=0):
        self.engines = engines
        self.sh = sharding
        self.K, self.N_total, self.d = K, N_total, d
        self.lag = int(append_lag)
        self.pending = []           # host exchange with lag: (visible_from_generation, rows)
        self.lib_exchange = sharding is not None and sharding.mode == "rccl" and sharding.world_size > 1
        self.peer_group = sharding is not None and sharding.mode == "peer"
        if self.peer_group and (sharding.world_size != 1 or self.lag):
            raise ValueError("mode='peer' groups the shards of ONE process and runs with append_lag 0")
        self.host_exchange = sharding is not None and not self.lib_exchange and not self.peer_group and (
            sharding.total_shards > 1 or sharding.host_exchange_always)
        if self.host_exchange:
            for e in engines:
                e.set_external_append(True)
        elif self.lag:
            for e in engines:
                e.set_append_lag(self.lag)      # the library applies the same visibility rule

    # generation loop --------------------------------------------------------------------------
    @property
    def library_loop(self):
        """True when the generation loop + its R-hat test can run as one library call (demcz_run_checked):
        one engine in this process and no host-driven exchange (single GPU, or RCCL inside the library)."""
        return len(self.engines) == 1 and not self.host_exchange and hasattr(self.engines[0], "run_checked")

    def run_checked(self, g_from, g_to, gamma, every, threshold=0.0, temperature=None):
        """demcz.jl:30-55 for a slab: returns (g_stop, [max R-hat per check], R-hat vector of the last check)."""
        if self.library_loop:
            return self.engines[0].run_checked(g_from, g_to, gamma, every, threshold, temperature)
        g, trace, last = g_from, [], None
        while g <= g_to:
            nxt = min(g_to, ((g - 1) // every + 1) * every)
            self.run(g, nxt, gamma, None if temperature is None else temperature[g - g_from:nxt - g_from + 1])
            if nxt % every == 0 and nxt >= every:
                last = self.rhat(nxt - every + 1, nxt)
                trace.append(float(np.max(last)))
                if threshold > 0 and trace[-1] < threshold:
                    return nxt, np.array(trace), last
            g = nxt + 1
        return g_to, np.array(trace), last

    def run(self, g_from, g_to, gamma, temperature=None):
        if not self.host_exchange:
            for e in self.engines:
                e.run(g_from, g_to, gamma, temperature)
            return
        g = g_from
        while g <= g_to:
            w_end = min(((g - 1) // self.K + 1) * self.K, g_to)
            self._admit(g)
            t = None if temperature is None else temperature[g - g_from:w_end - g_from + 1]
            for e in self.engines:
                e.run(g, w_end, gamma, t)
            if w_end % self.K == 0:
                rows = np.concatenate([e.get_state(with_Z=False)[0] for e in self.engines], axis=0)
                if self.sh.world_size > 1:
                    rows = np.concatenate(self.sh.all_gather(np.ascontiguousarray(rows)), axis=0)
                rows = np.asfortranarray(rows)
                if self.lag == 0:
                    for e in self.engines:
                        e.append_rows(rows)
                else:
                    # demcz_set_append_lag's rule: boundary j's batch closes at J = ceil(j/E)*E and its
                    # rows are drawn from generation (J + E)*K + 1 on
                    j = w_end // self.K
                    J = -(-j // self.lag) * self.lag
                    self.pending.append(((J + self.lag) * self.K + 1, rows))
            g = w_end + 1

    def _admit(self, g):
        while self.pending and self.pending[0][0] <= g:
            rows = self.pending.pop(0)[1]
            for e in self.engines:
                e.append_rows(rows)

    def flush(self):
        """Every appended row is in the archive afterwards (end of a run)."""
        self._admit(float("inf"))

    # statistics -------------------------------------------------------------------------------
    def _allsum(self, a):
        if self.sh is not None and self.sh.world_size > 1 and not self.lib_exchange:
            return self.sh.all_reduce_sum(np.ascontiguousarray(a))
        return a

    def rhat(self, g_from, g_to):
        """Rhat_gelman over all N_total chains, src/utils.jl:2-20."""
        if not self.host_exchange and len(self.engines) == 1:
            return self.engines[0].rhat(g_from, g_to)
        d = self.d
        n = (g_to - g_from + 1) // 2
        m = 2 * self.N_total
        s0 = self._allsum(sum(e.rhat_partial(g_from, g_to, 0, None) for e in self.engines))
        grand = s0 / m
        s1 = self._allsum(sum(e.rhat_partial(g_from, g_to, 1, grand) for e in self.engines))
        B = n / (m - 1) * s1[:d]
        W = s1[d:] / m
        varhat = (n - 1) / n * W + B / n
        return np.sqrt(varhat / W)

    def changed(self, g_from, g_to):
        c = sum(e.get_changed(g_from, g_to) for e in self.engines)
        if self.sh is not None and self.sh.world_size > 1:
            c = self.sh.all_reduce_sum(np.ascontiguousarray(c.astype(np.float64))).astype(np.int64)
        return c

    def changed_total(self, g_from, g_to):
        """sum(diff(log_obj[:, g_from-1:g_to], dims=2) .!= 0) over all chains: the count of demcz_anneal.jl:50, from the
        window kernels' ballot counters (no pass over the history)."""
        t = float(sum(e.changed_total(g_from, g_to) if hasattr(e, "changed_total") else int(np.sum(e.get_changed(g_from, g_to)))
                      for e in self.engines))
        if self.sh is not None and self.sh.world_size > 1:
            t = float(self.sh.all_reduce_sum(np.array([t]))[0])
        return int(t)

    def accept_ratio_mean(self, g_from, g_to):
        r = np.concatenate([e.accept_ratio(g_from, g_to) for e in self.engines])
        s = np.array([r.sum(), float(r.size)])
        if self.sh is not None and self.sh.world_size > 1:
            s = self.sh.all_reduce_sum(s)
        return s[0] / s[1]

    # results ----------------------------------------------------------------------------------
    def stream_history(self):
        """Single device engine: have the library stream every slab's history to pinned host mirrors while the next slab runs
        (demcz_history_stream); `history()` then returns arrays over those mirrors without copying.  Returns whether it is on."""
        e = self.engines[0]
        self._streamed = False
        if len(self.engines) == 1 and hasattr(e, "history_stream") and getattr(e, "Gcap", 0) > 0:
            try:
                e.history_stream(True)
                self._streamed = True
            except Exception:
                self._streamed = False
        return self._streamed

    def prepare_result_arrays(self, G, threads=4):
        """The arrays the history will come back into (mc.chain N x d x G, mc.log_obj N x G), made NOW and their pages
        touched by a few threads while the GPU runs: a fresh 0.5 GB array takes the download 20-28 ms of page faults, a
        touched one 9 ms (scripts/probes/pinned_copy.py).  Single engine, real device engine only; best effort."""
        self._res = None
        try:
            e = self.engines[0]
            if len(self.engines) != 1 or not hasattr(e, "_L") or G <= 0 or e.N * (e.d + 1) * G * 8 > (8 << 30):
                return
            import ctypes
            import threading
            ch = np.empty((e.N, e.d, G), order="F")
            lo = np.empty((e.N, G), order="F")
            ths = []
            for a in (ch, lo):
                n = a.nbytes
                step = -(-n // threads)
                for i in range(threads):
                    if i * step < n:      # (a foreign call: the GIL is released while it runs)
                        t = threading.Thread(target=ctypes.memset, args=(a.ctypes.data + i * step, 0, min(step, n - i * step)))
                        t.start()
                        ths.append(t)
            self._res = (ch, lo, ths)
        except Exception:
            self._res = None

    def _take_result_arrays(self):
        res, self._res = getattr(self, "_res", None), None
        if res is None:
            return None
        for t in res[2]:
            t.join()
        return res[0], res[1]

    def history(self, g_from, g_to, take=False):
        """`take`: the run is over and these are its result arrays -- a streamed history is handed over without a copy."""
        if take and len(self.engines) == 1 and getattr(self, "_streamed", False) and g_from == 1:
            self._streamed = False          # (the mirrors leave the handle with the arrays)
            return self.engines[0].take_history(g_from, g_to)
        if len(self.engines) == 1:          # (the engine's arrays are already column-major: no copy of 0.4 GB at C2)
            out = self._take_result_arrays()
            return self.engines[0].get_history(g_from, g_to, out=out) if out is not None else self.engines[0].get_history(g_from, g_to)
        parts = [e.get_history(g_from, g_to) for e in self.engines]
        n = sum(p[0].shape[0] for p in parts)
        chain = np.empty((n,) + parts[0][0].shape[1:], order="F")
        lobj = np.empty((n,) + parts[0][1].shape[1:], order="F")
        at = 0
        for c, l in parts:                  # one copy per shard, straight into its rows of the column-major result
            chain[at:at + c.shape[0]] = c
            lobj[at:at + c.shape[0]] = l
            at += c.shape[0]
        return chain, lobj

    def state(self):
        self.flush()
        sts = [e.get_state(with_Z=(i == 0)) for i, e in enumerate(self.engines)]
        X = np.asfortranarray(np.concatenate([s[0] for s in sts], axis=0))
        lp = np.concatenate([s[1] for s in sts])
        return X, lp, sts[0][2], sts[0][3]

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def close(self):
        for e in self.engines:
            e.close()


def make_runner(logobj, Zmat, N, K, Ngeneration, blockindex, eps_scale, X, logp, *, seed, sharding, device_id,
                 engine_factory, lanes_per_chain, stream, rng_offset=0, append_lag=0):
    M0, d = Zmat.shape
    if M0 < 2:
        raise ValueError("Zmat needs at least 2 rows (two distinct archive rows per proposal, demcz.jl:176-179)")
    sh = sharding
    shards = sh.total_shards if sh else 1
    if N % shards:
        raise ValueError("N must be divisible by the number of shards")
    n_loc = N // shards
    Mcap = M0 + int(math.ceil(N * Ngeneration / K))                     # demcz.jl:11
    nloc_shards = sh.local_shards if sh else 1
    first = (sh.rank * nloc_shards) if sh else 0
    factory = engine_factory or HipEngine
    engines = []
    for s in range(nloc_shards):
        c0 = (first + s) * n_loc
        e = factory(N=n_loc, d=d, K=K, Mcap=Mcap, Gcap=Ngeneration, blockindex=blockindex, eps_scale=eps_scale,
                    seed=seed, target=logobj, chain_id0=c0, device_id=device_id, stream=stream,
                    lanes_per_chain=lanes_per_chain)
        e.set_state(X[c0:c0 + n_loc], None if logp is None else logp[c0:c0 + n_loc], Zmat)
        if rng_offset:
            e.set_rng_offset(rng_offset)
        engines.append(e)
    if sh and sh.mode == "peer" and len(engines) > 1:
        type(engines[0]).peer_group(engines)
    if sh and sh.mode == "rccl" and sh.world_size > 1:
        uid = engines[0].comm_unique_id() if sh.rank == 0 else None
        uid = sh.broadcast_bytes(uid)
        engines[0].comm_init(uid, sh.world_size, sh.rank)
    return _Runner(engines, sh, K, N, d, append_lag=append_lag)


def blocks_per_generation(blockindex):
    """Philox blocks a chain consumes per generation: per block-step 1 (row indices) + ceil(nn / 2) (normal
    pairs; nn = b, or 1 for b = 1: demcz.jl:183-186) + 1 (the accept uniform)."""
    return sum(1 + ((1 if len(b) == 1 else len(b)) + 1) // 2 + 1 for b in blockindex)


def _generations_drawn(prevrun, blockindex=None):
    """Where a run that resumes `prevrun` starts its chains' random streams, in generations of ITS OWN size: the generations
    `prevrun` consumed (MC.rng_generations) -- scaled up to the next whole generation when the previous run's generations were
    of another size (other blocks), so that no draw is ever used twice."""
    n = getattr(prevrun, "rng_generations", None)
    n = int(prevrun.chain.shape[2]) if n is None else int(n)
    s_prev = getattr(prevrun, "rng_blocks_per_generation", None)
    if blockindex is not None and s_prev:
        s_new = blocks_per_generation([list(b) for b in blockindex])
        if s_new != int(s_prev):
            n = -(-n * int(s_prev) // s_new)
    return n


def initial_state(logobj, Zmat, N, Ngeneration, K, prevrun, init):
    """demcz.jl:13-22.  Returns (X, logp or None)."""
    M0, d = Zmat.shape
    if prevrun is None:
        if init == "last_rows":
            if M0 < N:
                raise ValueError("init='last_rows' needs at least N rows in Zmat")
            X = np.array(Zmat[M0 - N:M0, :], dtype=np.float64, order="F")
        elif init == "reference_zeros":
            # demcz.jl:11,15: the slice is taken from the zero-padded matrix
            pad = int(math.ceil(N * Ngeneration / K))
            padded = np.vstack([Zmat, np.zeros((pad, d))])
            X = np.array(padded[padded.shape[0] - N:, :], dtype=np.float64, order="F")
        else:
            raise ValueError("init must be 'last_rows' or 'reference_zeros'")
        logp = None
        if not is_device_target(logobj):
            logp = (np.asarray(logobj(X), dtype=np.float64).reshape(-1) if getattr(logobj, "batched", False)
                    else np.array([float(logobj(X[i, :].copy())) for i in range(N)]))      # demcz.jl:17
        return X, logp
    X = np.array(prevrun.chain[:, :, -1], dtype=np.float64, order="F")             # demcz.jl:20
    logp = np.array(prevrun.log_objcurrent, dtype=np.float64).reshape(-1)          # demcz.jl:21
    return X, logp


def _run_generations(runner, logobj, g_from, g_to, gamma, Nblocks, temperature=None):
    """Device targets: one call.  Python closures: the split propose / evaluate / commit loop."""
    if is_device_target(logobj):
        runner.run(g_from, g_to, gamma, temperature)
        return
    eng = runner.engines[0]
    # The proposals arrive in, and the closure's values leave from, pinned host buffers the kernels address directly (round 5:
    # demcz_closure_buffers) -- no copies, no stream synchronisation per block-step.  The reference's closure takes ONE parameter
    # vector (demcz.jl:189); a closure that declares `batched = True` is handed the whole N x d matrix of a block-step's
    # proposals and returns N values -- one call into NumPy instead of N.
    bufs = eng.closure_buffers() if hasattr(eng, "closure_buffers") else None
    batched = bool(getattr(logobj, "batched", False))
    for g in range(g_from, g_to + 1):
        T = None if temperature is None else float(temperature[g - g_from])
        for ib in range(Nblocks):
            Xp = eng.propose(g, ib, gamma)
            if bufs is not None:
                if batched:
                    bufs[1][:] = logobj(Xp)
                else:
                    for i in range(Xp.shape[0]):
                        bufs[1][i] = float(logobj(Xp[i, :].copy()))                       # demcz.jl:189
                eng.accept_commit(None, T)
            else:
                lp = (np.asarray(logobj(Xp), dtype=np.float64) if batched
                      else np.array([float(logobj(Xp[i, :].copy())) for i in range(Xp.shape[0])]))
                eng.accept_commit(lp, T)
        eng.end_generation(g)


def _warn_live_redos(runner):
    """A LIVE launch whose row hand-off timed out is redone with one launch per K-window (same results, several times slower,
    and the handle stays in that mode): worth a line in the caller's log -- typically another process shares the GPU."""
    for e in getattr(runner, "engines", []):
        st = getattr(e, "live_status", None)
        if st is None:
            continue
        try:
            _, redos = st()
        except Exception:
            continue
        if redos:
            warnings.warn(f"DEMCz: {redos} in-launch row hand-off(s) timed out and were redone with one launch per K-window "
                          "(results unchanged; is another process using this GPU?)", RuntimeWarning, stacklevel=3)
            return


def _finish(runner, prevrun, G, padded_Z, Mcap, rng_offset=0, blockindex=None):
    _warn_live_redos(runner)
    X, lp, Z, M = runner.state()               # (first: it only needs the compute stream; the history copies are still leaving)
    chain, lobj = runner.history(1, G, take=True)
    if padded_Z:
        Zp = np.zeros((Mcap, Z.shape[1]), order="F")
        Zp[:M] = Z
        Z = Zp
    if prevrun is not None:                                                          # demcz.jl:58-59
        chain = np.asfortranarray(np.concatenate([prevrun.chain, chain], axis=2))
        lobj = np.asfortranarray(np.concatenate([prevrun.log_obj, lobj], axis=1))
    spg = blocks_per_generation([list(b) for b in blockindex]) if blockindex is not None else None
    return MC(chain, lobj, X, lp, rng_generations=int(rng_offset) + int(G), rng_blocks_per_generation=spg), Z


def print_status(runner, ig, printlast=500):
    """src/demcz.jl:65-78"""
    lo, hi = max(1, ig - printlast), max(1, ig)
    chain, lobj = runner.history(lo, hi)
    print("-----------------------")
    print(f"iteration {ig}")
    print(f"average par = {chain.mean(axis=(0, 2))}")
    print(f"average val = {lobj.mean()}")
    print("-----------------------")


def print_status_anneal(runner, ig):
    """src/demcz_anneal.jl:5-12"""
    _, lobj = runner.history(1, ig)
    print("-----------------------")
    print(f"iteration {ig}")
    print(f"bestval = {lobj.max()}")
    print("-----------------------")


# --------------------------------------------------------------------------------------------
# demcz_sample (demcz.jl:1-63)
# --------------------------------------------------------------------------------------------
def demcz_sample(logobj, Zmat, N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=None, eps_scale=None, γ=2.38, *,
                 prevrun=None, verbose=True, print_step=100, autostop="no", autostop_Rhat=1.01,
                 autostop_every=1000, seed=0, init="last_rows", padded_Z=False, sharding=None, device_id=0,
                 lanes_per_chain=0, stream=None, engine_factory=None, return_runner=False, rng_offset=None,
                 append_lag=0):
    """Serial-driver surface of src/demcz.jl: pass a ``DEMCopt`` as the third argument
    (``demcz_sample(logobj, Zmat, opts; prevrun)``, :1-3) or the positional arguments with
    the positional method's defaults (:9).  Returns ``(mc, Z)``."""
    if isinstance(N, DEMCopt):
        o = N
        N, K, Ngeneration, Nblocks, blockindex, eps_scale, γ = o.N, o.K, o.Ngeneration, o.Nblocks, o.blockindex, o.eps_scale, o.γ
        verbose, print_step = o.verbose, o.print_step
        autostop, autostop_Rhat, autostop_every = o.autostop, o.autostop_Rhat, o.autostop_every
    Zmat = np.asarray(Zmat, dtype=np.float64)
    M0, d = Zmat.shape
    if blockindex is None:
        blockindex = [range(0, d)]
    if eps_scale is None:
        eps_scale = 1e-4 * np.ones(d)
    if len(blockindex) != Nblocks:
        raise ValueError("Nblocks != length(blockindex)")
    autostop = _sym(autostop)
    X, logp = initial_state(logobj, Zmat, N, Ngeneration, K, prevrun, init)
    if rng_offset is None:       # a resumed run continues the chains' random streams where prevrun stopped
        rng_offset = 0 if prevrun is None else _generations_drawn(prevrun, blockindex)
    runner = make_runner(logobj, Zmat, N, K, Ngeneration, blockindex, eps_scale, X, logp, seed=seed,
                          sharding=sharding, device_id=device_id, engine_factory=engine_factory,
                          lanes_per_chain=lanes_per_chain, stream=stream, rng_offset=rng_offset, append_lag=append_lag)
    if prevrun is None:
        if not (is_device_target(logobj) and runner.stream_history()):
            runner.prepare_result_arrays(Ngeneration)
    Mcap = M0 + int(math.ceil(N * Ngeneration / K))
    try:
        if verbose:
            print("-----------------------\niteration 0\n-----------------------")
        ig = 0
        if autostop == "Rhat" and not verbose and is_device_target(logobj) and prevrun is None:
            # the loop and its autostop test as one library call (demcz_run_checked): same decisions, no
            # host round trip per slab besides the R-hat read-back
            ig, _, _ = runner.run_checked(1, Ngeneration, γ, autostop_every, autostop_Rhat)
            if ig < Ngeneration or (ig % autostop_every == 0 and ig >= autostop_every and
                                    np.max(runner.rhat(ig - autostop_every + 1, ig)) < autostop_Rhat):
                if runner.accept_ratio_mean(ig - autostop_every + 1, ig) < 0.1:       # demcz.jl:42,44-46
                    print("Warning: accept ratio below 10% on average")
                res = _finish(runner, prevrun, ig, padded_Z, Mcap, rng_offset, blockindex)                  # demcz.jl:47-52
                return (res + (runner,)) if return_runner else res
        while ig < Ngeneration:                                                     # demcz.jl:30
            stops = [Ngeneration]
            if verbose:
                stops.append((ig // print_step + 1) * print_step)
            if autostop == "Rhat":
                stops.append((ig // autostop_every + 1) * autostop_every)
            nxt = min(stops)
            _run_generations(runner, logobj, ig + 1, nxt, γ, Nblocks)
            ig = nxt
            if verbose and ig % print_step == 0:                                    # demcz.jl:34-38
                print_status(runner, ig, printlast=autostop_every)
            if autostop == "Rhat" and ig % autostop_every == 0:                     # demcz.jl:39-40
                Rhat = runner.rhat(ig - autostop_every + 1, ig)                     # demcz.jl:41
                if np.max(Rhat) < autostop_Rhat:                                    # demcz.jl:43
                    # demcz.jl:42,44-46 (intent: per-chain ratio over log_obj[:, window])
                    if runner.accept_ratio_mean(ig - autostop_every + 1, ig) < 0.1:
                        print("Warning: accept ratio below 10% on average")
                    res = _finish(runner, prevrun, ig, padded_Z, Mcap, rng_offset, blockindex)              # demcz.jl:47-52
                    return (res + (runner,)) if return_runner else res
        res = _finish(runner, prevrun, Ngeneration, padded_Z, Mcap, rng_offset, blockindex)                 # demcz.jl:58-62
        return (res + (runner,)) if return_runner else res
    finally:
        if not return_runner:
            runner.close()


# --------------------------------------------------------------------------------------------
# demcz_anneal (demcz_anneal.jl:14-65)
# --------------------------------------------------------------------------------------------
def demcz_anneal(logobj, Zmat, N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=None, eps_scale=None, γ=2.38, *,
                 prevrun=None, verbose=True, print_step=100, temperaturefun=tempbaseline, T0=3, TN=0.0,
                 adaptγ=None, seed=0, init="last_rows", padded_Z=False, compat_serial_temp=False, sharding=None,
                 device_id=0, lanes_per_chain=0, stream=None, engine_factory=None, rng_offset=None, append_lag=0):
    """Simulated-annealing variant, src/demcz_anneal.jl:14-65: tempered accept
    ``log(u) < (lp' - lp)/T(ig)`` (:172-178), ``T(ig) = temperaturefun(ig, Ngeneration, T0, TN)``,
    gamma adapted from the windowed acceptance ratio every ``adapt_every`` generations (:48-57).

    ``compat_serial_temp=True`` reproduces the serial reference's actual schedule, which ignores
    T0/TN/Ngeneration and uses ``tempbaseline(ig, 1000, 1, 1e-3)``."""
    if isinstance(N, DEMCopt):
        o = N
        N, K, Ngeneration, Nblocks, blockindex, eps_scale, γ = o.N, o.K, o.Ngeneration, o.Nblocks, o.blockindex, o.eps_scale, o.γ
        verbose, print_step, T0, TN = o.verbose, o.print_step, o.T0, o.TN
    adapt = dict(DEFAULT_ADAPT)
    if adaptγ:
        adapt.update(adaptγ)
    Zmat = np.asarray(Zmat, dtype=np.float64)
    M0, d = Zmat.shape
    if blockindex is None:
        blockindex = [range(0, d)]
    if eps_scale is None:
        eps_scale = 1e-4 * np.ones(d)
    X, logp = initial_state(logobj, Zmat, N, Ngeneration, K, prevrun, init)
    if rng_offset is None:
        rng_offset = 0 if prevrun is None else _generations_drawn(prevrun, blockindex)
    runner = make_runner(logobj, Zmat, N, K, Ngeneration, blockindex, eps_scale, X, logp, seed=seed,
                          sharding=sharding, device_id=device_id, engine_factory=engine_factory,
                          lanes_per_chain=lanes_per_chain, stream=stream, rng_offset=rng_offset, append_lag=append_lag)
    if prevrun is None:
        if not (is_device_target(logobj) and runner.stream_history()):
            runner.prepare_result_arrays(Ngeneration)
    Mcap = M0 + int(math.ceil(N * Ngeneration / K))

    def temp(ig):
        if compat_serial_temp:
            return temperaturefun(ig, 1000, 1, 1e-3)                               # demcz_anneal.jl:67
        return temperaturefun(ig, Ngeneration, T0, TN)

    try:
        ae = int(adapt["adapt_every"])
        ig = 0
        while ig < Ngeneration:                                                     # demcz_anneal.jl:39
            stops = [Ngeneration]
            if verbose:
                stops.append((ig // print_step + 1) * print_step)
            if adapt["adapt"]:
                stops.append((ig // ae + 1) * ae)
            nxt = min(stops)
            temps = np.array([temp(g) for g in range(ig + 1, nxt + 1)], dtype=np.float64)
            _run_generations(runner, logobj, ig + 1, nxt, γ, Nblocks, temps)
            ig = nxt
            if verbose and ig % print_step == 0:
                print_status_anneal(runner, ig)                                     # demcz_anneal.jl:43-47
            if adapt["adapt"] and ig % ae == 0:                                     # demcz_anneal.jl:48-57
                # sum(diff(log_obj[:, ig-ae+1:ig], dims=2) .!= 0) / (N*ae): the first column of the
                # window has no predecessor inside the window
                nch = runner.changed_total(ig - ae + 2, ig) if ae > 1 else 0
                accept_ratio = float(nch) / (N * ae)
                if accept_ratio < 0.1:
                    γ = max(adapt["minγ"], γ * 0.5)
                elif accept_ratio > 0.5:
                    γ = min(adapt["maxγ"], γ * 1.5)
        return _finish(runner, prevrun, Ngeneration, padded_Z, Mcap, rng_offset, blockindex)                # demcz_anneal.jl:60-64
    finally:
        runner.close()
