"""Where a pass of the wave-per-chain kernel for d = 8 / 10 / 20 (demcz_kernels_pw.h) spends its time: shader-clock sums per
segment from a diagnostic build (-DDEMCZ_STAMPS, build_ab/stamps.so).  usage: python scripts/pw_stamps.py [d] [K] [generations]"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
so = ROOT / "build_ab" / "stamps.so"
os.environ["DEMCZ_LIB"] = str(so)          # (before anything imports demc_jl_amd: _lib reads it at import)
src = ROOT / "demc.jl_amd" / "csrc"
if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in src.glob("*")):
    so.parent.mkdir(exist_ok=True)
    sys.path.insert(0, str(ROOT))
    from demc_jl_amd import _lib as _build
    _build.build_lib(so, extra=["-DDEMCZ_STAMPS"])          # (every translation unit, in parallel: demc.jl_amd/_lib.py)
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc
from demc_jl_amd import _lib

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G = int(sys.argv[3]) if len(sys.argv) > 3 else 600
N = 1024
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G // 2, w["gamma"])
e.set_kernel_timing(True)
e.run(G // 2 + 1, G, w["gamma"])
nl, ms = e.get_kernel_time()
lib = _lib.load()
buf = np.zeros((N, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), N) == 0
s = buf.astype(np.float64)
assert np.all(s[:, 15] == 3), "the last launch was not window_kernel_pw's"
n = s[:, 14]
print(f"N={N} d={d} K={K}: {G - G // 2} generations in {nl} launches, {ms * 1e3:.1f} us (stamped build); last launch {n.mean():.0f} passes per chain wave")
print(f"  last launch: {s[:, 8].mean():.0f} shader clocks mean = {(s[:, 8] / n).mean():.0f} per pass")
names = ["state row + candidate adds straight from LDS", "history stores, DMA wait, next pass's increments, DMA issue", "table write, log-density (W through scalar loads)",
         "bpermute, accept tests, path", "history values, winner's row to row 0", "boundary", "waits for rows not yet published", "queue bookkeeping"]
for i, nm in enumerate(names):
    print(f"  {(s[:, i] / n).mean():8.0f} per pass  ({100 * s[:, i].sum() / s[:, 8].sum():5.1f} %)  {nm}")
