"""Pins the oracle's building blocks to published / independent references (CPU only).

The reference (chrished/DEMC.jl) holds no golden vectors for this path, so the oracle is pinned
piecewise: Philox4x32-10 against the Random123 known-answer vectors and against rocRAND's own
host-side stream, the spec's log / sincos against mpmath, the index draw against its definition.
"""
import shutil
import subprocess
from pathlib import Path

import mpmath as mp
import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

# Random123 kat_vectors, philox4x32 10 rounds: (counter, key) -> output
R123_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,out", R123_KAT)
def test_philox_random123_kat(oracle, ctr, key, out):
    assert oracle.philox(ctr, key) == out


def test_philox_stream_is_rocrand_stream(oracle):
    """oracle_draw_block(seed, chain, blk) == rocrand_init(seed, chain, 4*blk) + rocrand4()."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "rocrand_check"], check=True, stdout=subprocess.DEVNULL)
    exe = ROOT / "oracle" / "_build" / "rocrand_stream_dump"
    for seed, sub, blk in [(31953150, 0, 0), (31953150, 1023, 49999), (2**63 + 5, 2**33 + 7, 2**32 - 2)]:
        lines = subprocess.check_output([str(exe), str(seed), str(sub), str(blk), "5"]).decode().split("\n")
        for i in range(5):
            w = [int(x) for x in lines[i].split()]
            r1, r2 = oracle.draw_block(seed, sub, blk + i)
            assert [r1 & 0xffffffff, r1 >> 32, r2 & 0xffffffff, r2 >> 32] == w


def test_dm_log_accuracy(oracle):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(4000), rng.random(500) * 1e-12, 1 - rng.random(500) * 1e-9,
                         [2.0 ** -53, 1.5 * 2.0 ** -53, 1 - 2.0 ** -53, 0.5, 0.70710678118654752, 0.7071067811865476]])
    got = oracle.dm_log(xs)
    assert np.allclose(got, np.log(xs), rtol=4e-16, atol=0)
    mp.mp.prec = 120
    worst = 0.0
    for x, l in zip(xs[::7], got[::7]):
        t = mp.log(mp.mpf(float(x)))
        worst = max(worst, float(abs(mp.mpf(float(l)) - t) / mp.mpf(float(np.spacing(abs(float(t)))))))
    assert worst < 1.0, worst      # < 1 ulp


def test_dm_sincos2pi_accuracy(oracle):
    rng = np.random.default_rng(2)
    ks = np.concatenate([rng.integers(0, 2 ** 53, size=3000, dtype=np.uint64),
                         np.array([0, 1, 2 ** 50 - 1, 2 ** 50, 2 ** 50 + 1, 2 ** 51, 2 ** 52, 3 * 2 ** 51, 2 ** 53 - 1,
                                   2 ** 51 + 2 ** 50, 2 ** 52 + 2 ** 50], dtype=np.uint64)])
    cs = oracle.dm_sincos2pi(ks)
    mp.mp.prec = 120
    for k, (c, s) in zip(ks, cs):
        th = 2 * mp.pi * mp.mpf(int(k)) / 2 ** 53
        tc, ts = mp.cos(th), mp.sin(th)
        for got, true in ((c, tc), (s, ts)):
            err = abs(mp.mpf(float(got)) - true)
            assert err <= 2.3e-16                                   # absolute: <= 1 ulp of 1.0
            if abs(true) > 1e-3:
                assert err <= 2 * abs(true) * 2.0 ** -52            # relative: <= 2 ulp
    assert abs(cs[:, 0] ** 2 + cs[:, 1] ** 2 - 1).max() < 5e-16
    # exact quadrant values
    q = oracle.dm_sincos2pi(np.array([0, 2 ** 51, 2 ** 52, 3 * 2 ** 51], dtype=np.uint64))
    assert np.array_equal(np.abs(q), [[1, 0], [0, 1], [1, 0], [0, 1]])
    assert q[0, 0] == 1 and q[1, 1] == 1 and q[2, 0] == -1 and q[3, 1] == -1


def test_normal_pair_distribution(oracle):
    zs = []
    for blk in range(20000):
        r1, r2 = oracle.draw_block(7, 3, blk)
        zs.extend(oracle.normal_pair(r1, r2))
    zs = np.array(zs)
    n = zs.size
    assert abs(zs.mean()) < 4 / np.sqrt(n)
    assert abs(zs.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs(np.mean(zs ** 4) - 3) < 0.15
    assert abs(np.corrcoef(zs[0::2], zs[1::2])[0, 1]) < 4 / np.sqrt(n / 2)
    from scipy import stats
    assert stats.kstest(zs, "norm").pvalue > 1e-3


def test_index_draw_definition(oracle):
    """i1 = floor(r1 M / 2^64), j = floor(r2 (M-1) / 2^64), i2 = j + (j >= i1): two DISTINCT rows,
    uniform (the O(1) form of collect(1:M)/deleteat!, src/demcz.jl:176-179)."""
    d, N = 3, 1
    import demc_jl_amd as demc
    w = demc.workloads.mvnormal_problem(d, 8)
    for M in (2, 3, 50, 1000003):
        prob = oracle.Problem(N, d, 10, max(M, 8), w["eps_scale"], 5, target=w["target"].spec())
        Z = np.zeros((prob.Mcap, d), order="F")
        cnt = np.zeros((min(M, 50), min(M, 50)))
        for g in range(1, 400):
            out = oracle.block_step(prob, Z, M, 0, g, 0, 2.38, np.zeros(d), 0.0)
            r1, r2 = oracle.draw_block(5, 0, (g - 1) * prob.blocks_per_generation())
            i1 = (r1 * M) >> 64
            j = (r2 * (M - 1)) >> 64
            assert out["i1"] == i1 and out["i2"] == j + (1 if j >= i1 else 0)
            assert out["i1"] != out["i2"] and 0 <= out["i1"] < M and 0 <= out["i2"] < M
            if M <= 50:
                cnt[out["i1"], out["i2"]] += 1
        if M == 3:
            assert np.all(np.diag(cnt) == 0) and np.all(cnt[~np.eye(3, dtype=bool)] > 30)


def test_blocks_per_generation(oracle):
    import demc_jl_amd as demc
    w = demc.workloads.mvnormal_problem(6, 8)
    for blocks, expect in [([range(6)], 1 + 3 + 1), ([[0], [1, 2], [3, 4, 5]], (1 + 1 + 1) + (1 + 1 + 1) + (1 + 2 + 1)),
                           ([[5, 0, 3]], 1 + 2 + 1)]:
        prob = oracle.Problem(8, 6, 10, 100, w["eps_scale"], 1, blocks=blocks, target=w["target"].spec())
        assert prob.blocks_per_generation() == expect
