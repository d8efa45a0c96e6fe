"""ctypes binding of ``libdemcz_hip.so`` (C ABI: ``include/demcz.h``).

This is the only compute path of the package.  There is no CPU fallback: if the shared
object is missing, or no HIP device is visible, the calls raise -- they never degrade to a
host implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
# DEMCZ_LIB selects another build of the same HIP library (kernel A/B experiments); never a CPU path
LIB_PATH = Path(os.environ["DEMCZ_LIB"]) if os.environ.get("DEMCZ_LIB") else PKG_DIR / "libdemcz_hip.so"
ABI_VERSION = 1
# demcz_config.lanes_per_chain beyond 0 / 1 / 8 / 16 (include/demcz.h)
LAYOUT_SPLIT = 100          # producer / consumer split, eight replicated (d <= 10) or sixteen cooperating lanes per chain
LAYOUT_SPLIT_WAVE = 164     # the split with one wavefront per chain, five generations per pass (MvNormal, d = 2..5, 8, 10, 20)

TARGET_MVNORMAL, TARGET_ISO_QUAD, TARGET_LINREG_SSE, TARGET_HOST_CALLBACK = 0, 1, 2, 3
OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_CAPACITY, ERR_STATE, ERR_NO_DEVICE, ERR_COMM = range(7)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)

# every symbol include/demcz.h declares (tests check the library exports all of them)
SYMBOLS = [
    "demcz_abi_version", "demcz_create", "demcz_destroy", "demcz_last_error", "demcz_set_state",
    "demcz_get_state", "demcz_set_history_origin", "demcz_run", "demcz_synchronize",
    "demcz_get_history", "demcz_get_changed", "demcz_rhat", "demcz_accept_ratio", "demcz_mean_cov",
    "demcz_propose", "demcz_accept_commit", "demcz_end_generation", "demcz_comm_unique_id",
    "demcz_comm_init", "demcz_export_current_device", "demcz_append_rows_device",
    "demcz_set_external_append", "demcz_get_info", "demcz_selftest_draws", "demcz_append_rows",
    "demcz_rhat_partial", "demcz_set_rng_offset", "demcz_rhat_array", "demcz_accept_ratio_array",
    "demcz_mean_cov_array", "demcz_set_append_lag", "demcz_run_checked", "demcz_set_kernel_timing",
    "demcz_get_kernel_time", "demcz_set_live_spin_limit", "demcz_get_live_status",
    "demcz_debug_append_slab", "demcz_get_changed_total", "demcz_debug_set_live_fault",
    "demcz_set_comm_timeout", "demcz_debug_stall_exchange", "demcz_get_kernel_time_series",
    "demcz_history_stream", "demcz_get_history_view", "demcz_detach_history", "demcz_release_host_buffer", "demcz_get_archive_pinned",
    "demcz_debug_kernel_counts", "demcz_pool_trim", "demcz_debug_kernel_name", "demcz_peer_group", "demcz_get_peer_status", "demcz_peer_export", "demcz_peer_attach",
    "demcz_peer_detach", "demcz_get_peer_ping", "demcz_set_live_rearms", "demcz_get_live_rearms",
    "demcz_closure_buffers",
]


class DemczError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdemcz_hip status {code}: {msg}")
        self.code = code


class Config(C.Structure):
    """``demcz_config`` of include/demcz.h."""
    _fields_ = [
        ("N", C.c_int64), ("chain_id0", C.c_int64), ("d", C.c_int32), ("K", C.c_int32),
        ("Mcap", C.c_int64), ("Gcap", C.c_int64), ("Nblocks", C.c_int32),
        ("block_offsets", _ip), ("block_indices", _ip), ("eps_scale", _dp),
        ("seed", C.c_uint64), ("device_id", C.c_int32), ("target_kind", C.c_int32),
        ("mu", _dp), ("W", _dp), ("c0", C.c_double), ("design", _dp), ("yobs", _dp),
        ("nobs", C.c_int64), ("stream", C.c_void_p), ("lanes_per_chain", C.c_int32),
        ("reserved0", C.c_int32),
    ]


HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Werror", "-Wno-pass-failed"]


def sources() -> list:
    """The library's translation units: the C ABI with most kernels (demcz_capi.hip), the eight units that instantiate
    window_kernel_pw for every dimension from 6 to 32 (demcz_pw_inst_<g>.hip, csrc/demcz_pw_dispatch.h) and the two with the
    sixteen-lane regression kernels for every dimension from 2 to 28 (demcz_mlr_inst_<g>.hip, csrc/demcz_mlr_dispatch.h)."""
    csrc = PKG_DIR / "csrc"
    return [csrc / "demcz_capi.hip"] + sorted(csrc.glob("demcz_pw_inst_*.hip")) + sorted(csrc.glob("demcz_mlr_inst_*.hip"))


def build_command(out: Path = LIB_PATH) -> list:
    """The build as ONE command (what a Makefile-less integrator would type; compiles the units one after the other: ~4 min).
    build() below runs the same compiler with the same flags on every unit in parallel and links the objects."""
    return ["hipcc"] + HIPCC_FLAGS + ["-shared", "-o", str(out)] + [str(p) for p in sources()] + ["-lrccl"]


def build_lib(out: Path = LIB_PATH, extra: list = (), jobs: int = 0) -> Path:
    """hipcc -c per translation unit (in parallel: the units are independent), then one link.  `extra`: -D switches of an A/B build
    (scripts/build_variant.py).  Objects go to build/obj/<library name>/ (git-ignored)."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = REPO_ROOT / "build" / "obj" / Path(out).stem
    objdir.mkdir(parents=True, exist_ok=True)
    srcs = sources()
    jobs = jobs or max(1, min(len(srcs), len(os.sched_getaffinity(0))))

    def compile_one(src):
        obj = objdir / (src.stem + ".o")
        r = subprocess.run(["hipcc"] + list(extra) + HIPCC_FLAGS + ["-c", "-o", str(obj), str(src)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, srcs))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out)] + [str(o) for o in objs] + ["-lrccl"], check=True)
    return Path(out)


def build(force: bool = False) -> Path:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = list((PKG_DIR / "csrc").glob("*")) + [REPO_ROOT / "include" / "demcz.h"]
    if not force and LIB_PATH.exists():
        newest = max(p.stat().st_mtime for p in srcs)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    return build_lib(LIB_PATH)


_LIB = None


def load():
    """Load the library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    L.demcz_abi_version.restype = C.c_int32
    if L.demcz_abi_version() != ABI_VERSION:
        raise RuntimeError("libdemcz_hip.so ABI version mismatch: rebuild")
    L.demcz_last_error.restype = C.c_char_p
    L.demcz_last_error.argtypes = [C.c_void_p]
    L.demcz_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
    L.demcz_destroy.argtypes = [C.c_void_p]
    L.demcz_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_int64]
    L.demcz_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int64, _lp]
    L.demcz_set_history_origin.argtypes = [C.c_void_p, C.c_int64]
    L.demcz_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double, _dp]
    L.demcz_synchronize.argtypes = [C.c_void_p]
    L.demcz_get_history.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _dp, _dp]
    L.demcz_get_changed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _lp]
    L.demcz_get_changed_total.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _lp, _ip]
    L.demcz_rhat.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _dp]
    L.demcz_accept_ratio.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _dp]
    L.demcz_mean_cov.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _dp, _dp]
    L.demcz_propose.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_double, _dp]
    L.demcz_accept_commit.argtypes = [C.c_void_p, _dp, _dp]
    L.demcz_end_generation.argtypes = [C.c_void_p, C.c_int64]
    L.demcz_closure_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.demcz_comm_unique_id.argtypes = [C.c_void_p]
    L.demcz_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.demcz_export_current_device.argtypes = [C.c_void_p, C.c_void_p]
    L.demcz_append_rows_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    L.demcz_append_rows.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int64]
    L.demcz_rhat_partial.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, _dp, _dp]
    L.demcz_set_rng_offset.argtypes = [C.c_void_p, C.c_int64]
    L.demcz_set_append_lag.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_rhat_array.argtypes = [C.c_int32, _dp, C.c_int64, C.c_int32, C.c_int64, _dp]
    L.demcz_accept_ratio_array.argtypes = [C.c_int32, _dp, C.c_int64, C.c_int64, _dp]
    L.demcz_mean_cov_array.argtypes = [C.c_int32, _dp, C.c_int64, C.c_int32, C.c_int64, _dp, _dp]
    L.demcz_set_external_append.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_get_info.argtypes = [C.c_void_p, _lp, _lp, _ip]
    L.demcz_selftest_draws.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32,
                                       C.POINTER(C.c_uint64), _dp, _dp]
    L.demcz_set_live_spin_limit.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_get_live_status.argtypes = [C.c_void_p, _ip, _ip]
    L.demcz_debug_kernel_counts.argtypes = [C.c_void_p, _lp]
    L.demcz_history_stream.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_get_history_view.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.demcz_detach_history.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.demcz_release_host_buffer.argtypes = [C.c_void_p]
    L.demcz_pool_trim.argtypes = [_lp, _lp]
    L.demcz_peer_group.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    L.demcz_get_peer_status.argtypes = [C.c_void_p, _ip, _ip]
    L.demcz_peer_export.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.demcz_peer_attach.argtypes = [C.c_void_p, C.c_void_p]
    L.demcz_peer_detach.argtypes = [C.c_void_p]
    L.demcz_get_peer_ping.argtypes = [C.c_void_p, _ip, _dp]
    L.demcz_set_live_rearms.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_get_live_rearms.argtypes = [C.c_void_p, _ip, _ip]
    L.demcz_debug_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.demcz_get_archive_pinned.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _lp]
    L.demcz_get_kernel_time_series.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _ip]
    L.demcz_set_comm_timeout.argtypes = [C.c_void_p, C.c_int64]
    L.demcz_debug_stall_exchange.argtypes = [C.c_void_p, C.c_int32]
    L.demcz_debug_set_live_fault.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    L.demcz_debug_append_slab.argtypes = [C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_int32]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name != "demcz_last_error":
            fn.restype = C.c_int32
    _LIB = L
    return L


def f64(a, order="C"):
    return np.require(a, dtype=np.float64, requirements=["F" if order == "F" else "C", "A"])


def ptr(a, t=_dp):
    return a.ctypes.data_as(t) if a is not None else None
