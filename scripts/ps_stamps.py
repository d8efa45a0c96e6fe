"""What the in-launch row hand-off costs the wave-per-chain consumer (demcz_kernels_ps.h): shader-clock sums written by a
diagnostic build (-DDEMCZ_STAMPS, build_ab/stamps.so; never the shipped library).
usage: python scripts/ps_stamps.py [N] [K] [generations] [M0]   (run on the GPU box; M0: rows of a synthetic initial archive)"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
so = ROOT / "build_ab" / "stamps.so"
os.environ["DEMCZ_LIB"] = str(so)          # (before anything imports demc_jl_amd: _lib reads it at import)
if not so.exists():
    so.parent.mkdir(exist_ok=True)
    sys.path.insert(0, str(ROOT))
    from demc_jl_amd import _lib as _build
    _build.build_lib(so, extra=["-DDEMCZ_STAMPS"])          # (every translation unit, in parallel: demc.jl_amd/_lib.py)
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc
from demc_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
M0big = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = 5
w = demc.workloads.mvnormal_problem(d, N)
if M0big:
    w["Zinit"] = np.asfortranarray(w["mu"] + 0.1 * np.random.default_rng(0).standard_normal((M0big, d)))
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                   target=w["target"])
assert e.info()["lanes_per_chain"] == 164
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G // 2, 2.38)
e.run(G // 2 + 1, G, 2.38)
e.synchronize()
lib = _lib.load()
buf = np.zeros((N, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), N)
assert rc == 0, rc
s = buf.astype(np.float64)
n = s[:, 14]
tot, rr, ring, nrr, nring = s[:, 8], s[:, 9], s[:, 10], s[:, 11], s[:, 12]
print(f"N={N} K={K} M0={M0}: last launch, {n.mean():.0f} passes per chain wave; shader-clock ticks, mean / max over chain waves")
print(f"  {tot.mean():10.0f} {tot.max():10.0f}   whole launch ({(tot / n).mean():.0f} per pass)")
print(f"  {rr.mean():10.0f} {rr.max():10.0f}   waiting for rows other waves had not published yet "
      f"({100 * (nrr / n).mean():.2f} % of passes, {rr.sum() / max(nrr.sum(), 1):.0f} per wait)")
print(f"  {ring.mean():10.0f} {ring.max():10.0f}   waiting for the publisher to free a slot "
      f"({100 * (nring / np.maximum(n / 2, 1)).mean():.2f} % of boundaries)")
print(f"  share of the launch spent in those waits: {100 * ((rr + ring) / tot).mean():.1f} %")
e.close()
# who waits least is who sets the pace: by position in the workgroup (one of the four chain waves shares its SIMD with the
# publisher wave) and the spread over all chain waves
wt = rr / n
print("  waiting per pass by chain wave of the workgroup (0..3), mean ticks:", " ".join(f"{wt[k::4].mean():.0f}" for k in range(4)))
print(f"  waiting per pass over all chain waves: min {wt.min():.0f}  5 % {np.percentile(wt, 5):.0f}  median {np.median(wt):.0f}  95 % {np.percentile(wt, 95):.0f}  max {wt.max():.0f}")
lo = np.argsort(wt)[:12]
print("  the chain waves that wait least (index: ticks per pass):", ", ".join(f"{int(i)}: {wt[i]:.0f}" for i in lo))
