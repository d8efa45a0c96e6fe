"""Longer bit-equality runs of the wave-per-chain consumers (demcz_kernels_ps.h / _pw.h) against the one-lane fused kernel:
the same arithmetic spec through entirely different code -- LDS-DMA with counted waits, five generations per pass, the
in-launch row hand-off.  Timing-dependent faults (a DMA landing where LDS reads are still queued, a miscounted wait)
show up as ONE wrong accept in 10^5 passes, which the short parity cases never see; scripts/ps_stress.py is the long form."""
import numpy as np
import pytest

from helpers import SPLIT_WAVE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,N,K,tempered,lag", [(5, 1024, 10, False, 0), (5, 1000, 7, True, 0), (2, 513, 3, True, 0), (20, 1024, 10, False, 0),
                                                (20, 600, 7, True, 0), (5, 1024, 10, False, 3), (20, 1024, 10, False, 2), (8, 1024, 10, False, 0),
                                                (10, 700, 7, True, 0), (10, 1024, 10, False, 2)])
def test_wave_per_chain_equals_one_lane_kernel_over_many_passes(demc, d, N, K, tempered, lag):
    G = 1500
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    T = np.linspace(3.0, 0.5, G) if tempered else None
    res = {}
    for lanes in (SPLIT_WAVE, 1):
        e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"],
                           seed=11 + d, target=w["target"], lanes_per_chain=lanes)
        if lag:
            e.set_append_lag(lag)              # deferred visibility: one launch per `lag` K-windows, no in-launch hand-off
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        cuts = [0, G // 7, G // 2 + 3, G]
        for a, b in zip(cuts[:-1], cuts[1:]):
            e.run(a + 1, b, w["gamma"], None if T is None else T[a:b])
        ch, lo = e.get_history(1, G)
        X, lp, Z, M = e.get_state()
        res[lanes] = (ch, lo, X, lp, Z, e.changed_total(1, G))
        assert e.info()["lanes_per_chain"] == lanes
        assert e.live_status()[1] == 0
        e.close()
    a, b = res[SPLIT_WAVE], res[1]
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert a[5] == b[5]
