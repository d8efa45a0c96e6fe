"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs.
Bit-exact: the arithmetic spec makes every double identical."""
import numpy as np
import pytest

from helpers import oracle_sample

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


@pytest.mark.parametrize("N,d,G", [(4, 5, 200), (100, 5, 57), (1024, 5, 40), (64, 3, 30), (65, 8, 25), (32, 7, 25)])
def test_mvnormal_full_block_bit_exact(demc, oracle, N, d, G):
    w = demc.workloads.mvnormal_problem(d, N)
    seed = 99 + N
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], N, w["K"], G, 1, [range(d)], w["eps_scale"], w["gamma"],
                              verbose=False, seed=seed)
    ref = oracle_sample(oracle, w["target"], w["Zinit"], N, w["K"], G, None, w["eps_scale"], w["gamma"], seed)
    assert np.array_equal(mc.chain, ref["chain"])
    assert np.array_equal(mc.log_obj, ref["log_obj"])
    assert np.array_equal(mc.Xcurrent, ref["X"])
    assert np.array_equal(mc.log_objcurrent, ref["logp"])
    assert np.array_equal(Z, ref["Z"])


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()
