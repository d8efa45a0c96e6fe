"""Shared test helpers: run the oracle and the HIP path on the same seeded inputs."""
import numpy as np


def oracle_sample(O, target, Z0, N, K, G, blocks, eps, gamma, seed, temperature=None, schedule=0, init="last_rows",
                  X0=None, lp0=None, rng_offset=0, threads=0):
    """Oracle twin of demcz_sample's generation loop.  Returns dict(chain, log_obj, X, logp, Z, M, changed).
    threads > 0: the oracle's OpenMP loop over chains (synchronous schedule; the same bits as one thread)."""
    M0, d = Z0.shape
    Mcap = M0 + -(-N * G // K)
    prob = O.Problem(N, d, K, Mcap, eps, seed, blocks=blocks, target=target.spec())
    if X0 is None:
        X = np.array(Z0[M0 - N:], order="F") if init == "last_rows" else np.zeros((N, d), order="F")
    else:
        X = np.array(X0, order="F")
    lp = O.logp(prob, X) if lp0 is None else np.array(lp0, dtype=np.float64)
    Z = np.zeros((Mcap, d), order="F")
    Z[:M0] = Z0
    M, chain, lobj, changed = O.run(prob, X, lp, Z, M0, 1, G, gamma, temperature=temperature, schedule=schedule,
                                     rng_offset=rng_offset, threads=threads)
    return dict(chain=chain, log_obj=lobj, X=X, logp=lp, Z=Z[:M].copy(), M=M, changed=changed, prob=prob)


SPLIT, SPLIT_WAVE = 100, 164       # DEMCZ_LAYOUT_SPLIT, DEMCZ_LAYOUT_SPLIT_WAVE (include/demcz.h)


def auto_split_layout(d, N, K=10):
    """What lanes_per_chain = 0 selects for MvNormal with one full block: one wave per chain for the smallest populations of every
    d in 2..32 (as many chains as a LIVE launch of it holds: 1024 on MI355X; 2048 at d <= 5 with K a multiple of five, where a
    wave runs two chains), the replicated / cooperating consumers otherwise where they are built (d <= 10, d = 20)."""
    if 2 <= d <= 32 and N <= 1024:
        return SPLIT_WAVE
    if 2 <= d <= 5 and N <= 2048 and K % 5 == 0:
        return SPLIT_WAVE
    return SPLIT


def split_built(d):
    """MvNormal, one full block: dimensions with the eight-lane replicated (d <= 10) or sixteen-lane cooperating (d = 20) consumers."""
    return 2 <= d <= 10 or d == 20
