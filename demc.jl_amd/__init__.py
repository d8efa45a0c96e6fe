"""demc.jl_amd -- MI355X-native DEMCz chain-update engine behind the call surface of
chrished/DEMC.jl (``demcopt``, ``demcz_sample``, ``demcz_anneal``, ``MC``).

The directory name has a dot, so import it through the repo-root alias module::

    import demc_jl_amd as demc
    mc, Z = demc.demcz_sample(demc.MvNormalTarget(mu, Sigma), Zinit, demc.demcopt(5, N=1024))

All computation below the generation loop runs in ``libdemcz_hip.so`` (HIP, gfx950) through the
C ABI of ``include/demcz.h``; there is no CPU fallback.
"""
from ._lib import DemczError, build, LIB_PATH, SYMBOLS, LAYOUT_SPLIT, LAYOUT_SPLIT_WAVE          # noqa: F401
from .engine import HipEngine, selftest_draws, pool_trim                   # noqa: F401
from .targets import MvNormalTarget, IsoQuadTarget, LinRegSSETarget, is_device_target   # noqa: F401
from .sampler import (MC, DEMCopt, demcopt, demcz_sample, demcz_anneal, tempbaseline,   # noqa: F401
                      make_runner, initial_state,
                      Sharding, DEFAULT_ADAPT)
from .utils import (Rhat_gelman, flatten_chain, accept_ratio, mean_cov_chain, convergence_check,   # noqa: F401
                    save_checkpoint, load_checkpoint)
from . import workloads                                         # noqa: F401
