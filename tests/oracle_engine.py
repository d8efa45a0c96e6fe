"""TEST-ONLY engine with HipEngine's methods, backed by the CPU oracle.

Lets the host logic of demc.jl_amd/sampler.py (slab loop, autostop, gamma adaptation, sharded
exchange) run on machines without a GPU, e.g. in the world_size-2 gloo tests.  The product never
constructs this class; it lives under tests/ and is injected through ``engine_factory=``.
"""
import numpy as np

import oracle_py as O


class OracleEngine:
    def __init__(self, *, N, d, K, Mcap, Gcap, blockindex, eps_scale, seed, target, chain_id0=0, device_id=0,
                 stream=None, lanes_per_chain=0, threads=0):
        self.threads = int(threads)        # > 0: the oracle's OpenMP loop over chains (same bits; for the long full-size cases)
        self.N, self.d, self.K, self.Mcap, self.Gcap = N, d, K, Mcap, Gcap
        self.prob = O.Problem(N, d, K, Mcap, eps_scale, seed, blocks=[list(b) for b in blockindex],
                              chain_id0=chain_id0, target=target.spec())
        self.chain = np.zeros((N, d, Gcap), order="F")
        self.log_obj = np.zeros((N, Gcap), order="F")
        self.changed = np.zeros(Gcap, dtype=np.int64)
        self.external = False
        self.M = 0
        self.rng_offset = 0

    def set_rng_offset(self, generations):
        self.rng_offset = int(generations)

    def set_state(self, X, logp, Z):
        self.X = np.array(X, dtype=np.float64, order="F")
        self.Z = np.zeros((self.Mcap, self.d), order="F")
        self.Z[:Z.shape[0]] = Z
        self.M = Z.shape[0]
        self.lp = O.logp(self.prob, self.X) if logp is None else np.array(logp, dtype=np.float64)

    def set_external_append(self, enabled):
        self.external = bool(enabled)

    def run(self, g_from, g_to, gamma, temperature=None):
        M, ch, lo, cg = O.run(self.prob, self.X, self.lp, self.Z, self.M, g_from, g_to, gamma,
                              temperature=temperature, do_append=not self.external, rng_offset=self.rng_offset, threads=self.threads)
        self.M = M
        self.chain[:, :, g_from - 1:g_to] = ch
        self.log_obj[:, g_from - 1:g_to] = lo
        self.changed[g_from - 1:g_to] = cg

    def append_rows(self, rows):
        n = rows.shape[0]
        self.Z[self.M:self.M + n] = rows
        self.M += n

    def get_state(self, with_Z=True):
        return self.X.copy(order="F"), self.lp.copy(), (self.Z[:self.M].copy(order="F") if with_Z else None), self.M

    def get_history(self, g_from, g_to, chain=True, log_obj=True):
        return (np.asfortranarray(self.chain[:, :, g_from - 1:g_to]), np.asfortranarray(self.log_obj[:, g_from - 1:g_to]))

    def get_changed(self, g_from, g_to):
        return self.changed[g_from - 1:g_to].copy()

    def changed_total(self, g_from, g_to):
        return int(self.changed[g_from - 1:g_to].sum())

    def rhat(self, g_from, g_to):
        return O.rhat_gelman(self.chain[:, :, g_from - 1:g_to])

    def rhat_partial(self, g_from, g_to, stage, grand):
        w = g_to - g_from + 1
        n = w // 2
        win = self.chain[:, :, g_from - 1:g_from - 1 + 2 * n]
        halves = np.concatenate([win[:, :, :n], win[:, :, n:]], axis=0)      # (2N, d, n)
        mean_j = halves.mean(axis=2)
        if stage == 0:
            return mean_j.sum(axis=0)
        s2 = ((halves - mean_j[:, :, None]) ** 2).sum(axis=2) / (n - 1)
        return np.concatenate([((mean_j - grand[None, :]) ** 2).sum(axis=0), s2.sum(axis=0)])

    def accept_ratio(self, g_from, g_to):
        lo = self.log_obj[:, g_from - 1:g_to]
        return O.changed_per_chain(lo) / (lo.shape[1] - 1)

    def synchronize(self):
        pass

    def close(self):
        pass
