"""Post-hoc diagnostics with the reference's names (src/utils.jl), reduced on the GPU.

``Rhat_gelman``, ``mean_cov_chain`` and ``convergence_check`` take the host arrays the reference's
examples pass them (test/example_normpdf.jl:35-47); the arrays are uploaded and reduced by the same
kernels the autostop uses.  ``flatten_chain`` is a pure re-indexing (utils.jl:22-32).  Plotting,
``save_res`` and ``extract_best`` are not reproduced (dead code in Julia >= 1.0).
Checkpoints: the reference only resumes in memory (``prevrun=``); ``save_checkpoint`` /
``load_checkpoint`` put the same information in one ``.npz`` file.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DemczError
from .sampler import MC


def _chk(rc):
    if rc != 0:
        raise DemczError(rc, (_lib.load().demcz_last_error(None) or b"").decode())


def Rhat_gelman(chain, Npop=None, Ngeneration=None, Npar=None, device_id=0):
    """Split-chain Gelman-Rubin statistic per parameter, src/utils.jl:2-20."""
    chain = _lib.f64(chain, "F")
    N, d, G = chain.shape
    if (Npop, Ngeneration, Npar) != (None, None, None):
        assert (Npop or N, Npar or d) == (N, d) and (Ngeneration or G) <= G
        G = Ngeneration or G
        chain = _lib.f64(chain[:, :, :G], "F")
    out = np.empty(d)
    _chk(_lib.load().demcz_rhat_array(device_id, _lib.ptr(chain), N, d, G, _lib.ptr(out)))
    return out


def flatten_chain(chain, Npop=None, Ngeneration=None, Npar=None):
    """Npar x (Npop*Ngeneration) matrix, generation-major then chain, src/utils.jl:22-32."""
    chain = np.asarray(chain)
    N, d, G = chain.shape
    return chain.transpose(1, 2, 0).reshape(d, G * N)


def accept_ratio(log_obj, device_id=0):
    """sum(diff(log_obj, dims=2) .!= 0, dims=2) ./ (Ngeneration-1), src/utils.jl:61."""
    log_obj = _lib.f64(log_obj, "F")
    N, G = log_obj.shape
    out = np.empty(N)
    _chk(_lib.load().demcz_accept_ratio_array(device_id, _lib.ptr(log_obj), N, G, _lib.ptr(out)))
    return out


def mean_cov_chain(chain, Npop=None, Ngeneration=None, Npar=None, device_id=0):
    """Mean over all Npop*Ngeneration draws and their 1/(Npop*Ngeneration) covariance, src/utils.jl:96-111."""
    chain = _lib.f64(chain, "F")
    N, d, G = chain.shape
    mean = np.empty(d)
    cov = np.empty((d, d), order="F")
    _chk(_lib.load().demcz_mean_cov_array(device_id, _lib.ptr(chain), N, d, G, _lib.ptr(mean), _lib.ptr(cov)))
    return mean, cov


def convergence_check(chain, log_obj, figure_path=None, verbose=True, parnames=None, device_id=0):
    """(accept_ratio, Rhat) as src/utils.jl:34-94 returns them (the plotting part is commented out
    in the reference and is not reproduced)."""
    chain = np.asarray(chain)
    log_obj = np.asarray(log_obj)
    Npop, Npar, Ngeneration = chain.shape
    if log_obj.shape != (Npop, Ngeneration):
        raise ValueError("log_obj must be Npop x Ngeneration")                         # utils.jl:40-46
    acc = accept_ratio(log_obj, device_id)
    Rhat = Rhat_gelman(chain, device_id=device_id)
    if verbose:
        print("Summary Checks\n\nAcceptance Ratio of each chain:")
        print(acc)
        print(f"\nRhat Gelman: {Rhat}\n")
    return acc, Rhat


_NO_SEED = object()


def save_checkpoint(path, mc: MC, Z, generations_done=None, seed=_NO_SEED, opts=None):
    """Everything a resumed run needs: final states, archive, how far the RNG streams have advanced
    (`generations_done`; default: what `mc` itself records, MC.generations_drawn) and the `seed` of the run -- REQUIRED
    (positionally or by keyword): a checkpoint written with a wrong seed resumes on another Philox stream, silently."""
    if seed is _NO_SEED:
        raise TypeError("save_checkpoint: pass the run's seed (the one given to demcz_sample / demcz_anneal)")
    if generations_done is None:
        generations_done = mc.generations_drawn
    np.savez_compressed(path, Xcurrent=mc.Xcurrent, log_objcurrent=mc.log_objcurrent, last_chain=mc.chain[:, :, -1:],
                        last_log_obj=mc.log_obj[:, -1:], Z=Z, generations_done=int(generations_done), seed=int(seed),
                        blocks_per_generation=int(getattr(mc, "rng_blocks_per_generation", None) or 0))


def load_checkpoint(path):
    """Returns (prevrun, Z, generations_done, seed): pass ``prevrun=prevrun, seed=seed`` to ``demcz_sample`` to continue
    the same streams -- the returned ``prevrun`` keeps only the last generation of the history but records
    ``generations_done`` (MC.rng_generations), so the resumed run draws what an uninterrupted one would."""
    f = np.load(path)
    prev = MC(np.asfortranarray(f["last_chain"]), np.asfortranarray(f["last_log_obj"]), np.asfortranarray(f["Xcurrent"]),
              np.array(f["log_objcurrent"]), rng_generations=int(f["generations_done"]),
              rng_blocks_per_generation=(int(f["blocks_per_generation"]) or None) if "blocks_per_generation" in f else None)
    return prev, np.asfortranarray(f["Z"]), int(f["generations_done"]), int(f["seed"])
