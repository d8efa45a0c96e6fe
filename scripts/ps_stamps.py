"""Where a pass of the wave-per-chain consumer (demcz_kernels_ps.h) spends its time: shader-clock sums written by a
diagnostic build (-DDEMCZ_STAMPS, build_ab/stamps.so; never the shipped library).
usage: python scripts/ps_stamps.py [N] [K] [generations] [M0]   (run on the GPU box; M0: rows of a synthetic initial archive)"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
so = ROOT / "build_ab" / "stamps.so"
if not so.exists():
    so.parent.mkdir(exist_ok=True)
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-pass-failed",
                    "-DDEMCZ_STAMPS", "-o", str(so), str(ROOT / "demc.jl_amd" / "csrc" / "demcz_capi.hip"), "-lrccl"], check=True)
os.environ["DEMCZ_LIB"] = str(so)
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
names = ["candidates: state + the node's rows (and whatever the loop's top waits for)",
         "previous pass's history stores; front end of the next pass: slot wait, increments -> LDS, DMA, node rows <- LDS",
         "log-density of all nodes, candidates -> LDS", "accept tests, path from the lane mask, new state from the winner's registers",
         "history values <- LDS, boundary (append hand-off)", "LIVE re-reads, bookkeeping of the passes"]
print(f"N={N} K={K}: last launch, {n.mean():.0f} passes per chain wave; shader-clock ticks per pass, mean / max over chains")
tot = 0
for i, nm in enumerate(names):
    v = s[:, 8 + i] / n
    tot += v.mean()
    print(f"  {v.mean():8.0f} {v.max():8.0f}   {nm}")
print(f"  {tot:8.0f}            per pass in all")
print(f"  passes that had to ask again for a row: {100 * (s[:, 15] / n).mean():.2f} %")
e.close()
