"""Where a block-step of the block-update split kernel (window_kernel_mlb, C3) goes: shader-clock sums per wave from a
-DDEMCZ_STAMPS build.  usage: python scripts/mlb_stamps.py [N] [gens]   (DEMCZ_NO_LIVE=1: one launch per K-window, no in-launch hand-off)"""
import ctypes as C, os, subprocess, sys
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 102
w = demc.workloads.mvnormal_problem(d, N)
blocks = [range(0, 5), range(5, 10), range(10, 15), range(15, 20)]
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=blocks, eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G, w["gamma"]); e.synchronize()
lib = _lib.load()
nw = N // 4
buf = np.zeros((nw, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), nw) == 0
s = buf.astype(np.float64)
ns = s[:, 14]
names = ["wait for the step's rows, sentinel poll, increments", "next block-step's draws: entry -> LDS -> rows asked for", "dependent part: r -> W r -> q -> accept",
         "history row, append", "-", "between steps"]
print(f"N={N} gens={G} archive {e.M} rows: last launch {int(ns[0])} block-steps per wave; ticks per block-step, mean over waves / max")
for i in (0, 1, 2, 3, 5):
    v = s[:, 8 + i] / ns
    print(f"  {v.mean():8.0f} {v.max():8.0f}   {names[i]}")
print(f"  {(s[:, 8:14].sum(axis=1) / ns).mean():8.0f} ticks per block-step in all")
e.close()
