"""Oracle log-densities against independent closed forms (scipy / numpy)."""
import numpy as np
import pytest
from scipy import stats


@pytest.mark.parametrize("d", [1, 2, 5, 10, 20])
def test_mvnormal_logpdf_vs_scipy(oracle, demc, d):
    """logpdf(MvNormal(mu, Sigma), x), test/example_normpdf.jl:13-16 (Distributions.jl is not
    vendored in the reference; scipy's closed form is the independent check)."""
    w = demc.workloads.mvnormal_problem(d, 8)
    prob = oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    rng = np.random.default_rng(d)
    X = w["mu"] + rng.standard_normal((200, d)) * 0.2
    got = oracle.logp(prob, X)
    ref = stats.multivariate_normal(w["mu"], w["Sigma"]).logpdf(X)
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-9)
    t = w["target"]
    assert np.allclose(t.W @ w["Sigma"] @ t.W.T, np.eye(d), atol=1e-10)
    assert np.allclose(np.triu(t.W, 1), 0)


def test_iso_quad_and_linreg(oracle, demc):
    w = demc.workloads.iso_quad_problem(10, 8)
    prob = oracle.Problem(8, 10, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    X = np.random.default_rng(0).standard_normal((50, 10))
    assert np.allclose(oracle.logp(prob, X), -((X - w["mu"]) ** 2).sum(axis=1), rtol=1e-13)
    w = demc.workloads.linreg_problem(10, 8, nobs=300)
    prob = oracle.Problem(8, 10, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    B = w["beta"] + 0.1 * np.random.default_rng(1).standard_normal((40, 10))
    ref = np.array([-0.5 * np.sum((w["y"] - w["design"] @ b) ** 2) for b in B])
    assert np.allclose(oracle.logp(prob, B), ref, rtol=1e-12)


def _grouped_logp_python(t, x, offs):
    """DESIGN.md section 3, "grouped by the blocks", written out with Python floats and math.fma-free exact steps is not
    possible (no fma in Python 3.10): the restatement below uses numpy longdouble products rounded ONCE per fma, which is what
    an fma is for operands whose product fits 80-bit precision's 64-bit significand only approximately -- so it is compared at
    1 ulp, and the GROUPING (which terms go into which partial sum, in which order) is what it pins."""
    d = len(t.mu)
    r = np.asarray(x, dtype=np.float64) - t.mu
    W = np.asarray(t.W)

    def fma(a, b, c):
        return float(np.longdouble(a) * np.longdouble(b) + np.longdouble(c))
    ng = len(offs) - 1
    q = 0.0
    for g in range(ng):
        Qg = 0.0
        for i in range(offs[g], offs[g + 1]):
            y = 0.0
            for gb in range(g + 1):
                jl, jh = offs[gb], min(offs[gb + 1] - 1, i)
                P = float(W[i, jl] * r[jl])
                for j in range(jl + 1, jh + 1):
                    P = fma(W[i, j], r[j], P)
                y = P if gb == 0 else y + P
            Qg = y * y if i == offs[g] else fma(y, y, Qg)
        q = Qg if g == 0 else q + Qg
    return fma(-0.5, q, t.c0)


@pytest.mark.parametrize("d,blocks", [(20, [list(range(0, 5)), list(range(5, 10)), list(range(10, 15)), list(range(15, 20))]),
                                      (6, [[0], [1, 2], [5, 3, 4]]), (5, [[1, 0], [2], [4, 3]]), (10, [list(range(0, 7)), [7, 8, 9]])])
def test_mvnormal_sums_grouped_by_consecutive_blocks(oracle, demc, d, blocks):
    """Round 4: with blocks that are consecutive index ranges the oracle cuts the quadratic form's sums at the block boundaries
    (oracle/demcz_oracle.c: mvn_groups, target_logp).  The grouped value follows the written-out order to the last bit or one,
    agrees with the one-group order (and scipy) to rounding, and differs from it in the last bits somewhere -- it IS another order."""
    w = demc.workloads.mvnormal_problem(d, 8)
    t = w["target"]
    grouped = oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, blocks=blocks, target=t.spec())
    plain = oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, target=t.spec())
    rng = np.random.default_rng(100 + d)
    X = w["mu"] + rng.standard_normal((300, d)) * 0.2
    a, b = oracle.logp(grouped, X), oracle.logp(plain, X)
    assert np.allclose(a, b, rtol=1e-13, atol=1e-11)
    assert np.any(a != b), "the grouped order must be a different summation order"
    offs = [0] + list(np.cumsum([len(bl) for bl in blocks]))
    ref = np.array([_grouped_logp_python(t, x, offs) for x in X[:60]])
    assert np.all(np.abs(a[:60] - ref) <= 2 * np.spacing(np.abs(ref))), "grouping differs from the written-out order"
    assert np.allclose(a, stats.multivariate_normal(w["mu"], w["Sigma"]).logpdf(X), rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("blocks", [[[3], [0], [4], [1], [2]], [[0, 2], [1, 3, 4]], [[0, 1, 2]], [[0, 1], [1, 2, 3, 4]], [[0, 1, 2, 3, 4]]])
def test_mvnormal_sums_stay_in_one_group_otherwise(oracle, demc, blocks):
    """Blocks that are not consecutive ranges in order (permuted, interleaved, not covering, overlapping) or a single block keep
    the order of rounds 1-3: bit-equal to the run without block structure."""
    d = 5
    w = demc.workloads.mvnormal_problem(d, 8)
    X = w["mu"] + np.random.default_rng(3).standard_normal((100, d)) * 0.3
    a = oracle.logp(oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, blocks=blocks, target=w["target"].spec()), X)
    b = oracle.logp(oracle.Problem(8, d, 10, 100, w["eps_scale"], 1, target=w["target"].spec()), X)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("d,blocks", [(6, [[0], [1, 2], [5, 3, 4]]), (6, [[0, 1, 2], [3, 4, 5]]),
                                      (20, [list(range(0, 5)), list(range(5, 10)), list(range(10, 15)), list(range(15, 20))]),
                                      (20, [list(range(0, 10)), list(range(10, 13)), list(range(13, 20))])])
def test_grouped_mvnormal_order_stays_within_ulps_of_the_ungrouped_one(oracle, d, blocks):
    """when a run's blocks are consecutive index ranges, the MvNormal quadratic form is summed group by group
    (oracle/demcz_oracle.c: mvn_groups), so the same point x gets different LAST BITS under Nblocks = 1 and Nblocks = 4 -- the
    reference has no canonical order for a sum of d products (Distributions / PDMats: not under /root/reference).  How far apart:
    at most 8 units in the last place of the larger operand of the final fma, max(|c0|, q/2) -- measured 6 over 3 x 10^5 points
    from 0.01 to 10 standard scales away from the mean (relative to log p itself the figure is unbounded where c0 and q/2 cancel).
    What that means for a caller: a `prevrun` / a host closure built for other blocks carries log-densities that differ from this
    run's own evaluation in the 15th-16th significant digit -- the size of the accept test's own rounding, not of anything a
    Metropolis decision turns on -- and tests/test_gpu_parity.py checks that the HIP path and the oracle treat such carried values
    identically."""
    import demc_jl_amd as demc
    w = demc.workloads.mvnormal_problem(d, 64)
    one = oracle.Problem(64, d, 10, 100, w["eps_scale"], 1, target=w["target"].spec())
    grp = oracle.Problem(64, d, 10, 100, w["eps_scale"], 1, blocks=blocks, target=w["target"].spec())
    c0 = float(oracle.logp(one, w["mu"][None, :])[0])
    rng = np.random.default_rng(5)
    worst, differing = 0.0, 0
    for scale in (0.01, 0.03, 0.1, 0.3, 1.0, 10.0):
        X = w["mu"] + scale * rng.standard_normal((50000, d))
        a, b = oracle.logp(one, X), oracle.logp(grp, X)
        mag = np.maximum(abs(c0), np.abs(c0 - a))
        worst = max(worst, float((np.abs(a - b) / np.spacing(mag)).max()))
        differing += int(np.count_nonzero(a != b))
    assert differing > 0, "the two orders are different arithmetic: some points must differ in their last bits"
    assert worst <= 8.0, worst
