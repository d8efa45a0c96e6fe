"""Replicated consumer (window_kernel_pc8): is any chain wave slower than the rest all the time (two chain waves on one SIMD)?
Shader clocks of the state-dependent part of a chunk ("generations done"), per workgroup, from a -DDEMCZ_STAMPS build.
usage: python scripts/pc8_wave_speed.py [N] [gens]"""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["DEMCZ_LIB"] = str(ROOT / "build_ab" / "stamps.so")
os.environ["DEMCZ_NO_PS"] = "1"
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc
from demc_jl_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
d, K = 5, 10
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G, 2.38); e.synchronize()
lib = _lib.load()
ncons = (N + 7) // 8
buf = np.zeros((ncons, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), ncons) == 0
c = buf.astype(np.float64)
nch = c[:, 14]
for j, nm in ((8, "rows there, increments (waits included)"), (9, "next chunk's loads issued"), (10, "generations done"), (11, "append + hand-off")):
    v = c[:, j] / nch
    print(f"  {nm:42s} min {v.min():7.0f}  5 % {np.percentile(v, 5):7.0f}  median {np.median(v):7.0f}  95 % {np.percentile(v, 95):7.0f}  max {v.max():7.0f}")
v = c[:, 10] / nch
print(f"N={N}: {ncons} chain waves, {nch.mean():.0f} chunks; chain waves whose state-dependent part is > 1.1 x median: {(v > 1.1 * np.median(v)).sum()}")
print(f"lanes {e.info()['lanes_per_chain']} live {e.live_status()}")
e.close()
