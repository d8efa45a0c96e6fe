"""Where a block-step of the block-update split kernel (window_kernel_mlb, C3) goes: shader-clock sums per wave from a
-DDEMCZ_STAMPS build.  usage: python scripts/mlb_stamps.py [N] [gens]   (DEMCZ_NO_LIVE=1: one launch per K-window, no in-launch hand-off)"""
import ctypes as C, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
so = ROOT / "build_ab" / (os.environ.get("DEMCZ_STAMPS_LIB", "stamps") + ".so")      # (DEMCZ_STAMPS_LIB=stamps_nohist: an experiment build)
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
names = ["waiting for rows not yet published (sentinel poll), increments", "next block-step's draws: entry -> LDS -> rows asked for", "dependent part: r -> W r -> q -> accept",
         "history row, append", "the step's rows and record arrive (asked for a block-step ago)", "between steps"]
print(f"N={N} gens={G} archive {e.M} rows: last launch {int(ns[0])} block-steps per wave; ticks per block-step, mean over waves / max")
for i in (4, 0, 1, 2, 3, 5):
    v = s[:, 8 + i] / ns
    print(f"  {v.mean():8.0f} {v.max():8.0f}   {names[i]}")
print(f"  {(s[:, 8:14].sum(axis=1) / ns).mean():8.0f} ticks per block-step in all")
print(f"  {(s[:, 6] / ns).mean():8.0f} {(s[:, 6] / ns).max():8.0f}   of which INSIDE the poll loop (round 5: timed only in the steps that enter it -- a stamp costs 100-200 clocks, "
      "which round 4's figure for the waits, this segment's total divided by the share of steps with a wait, attributed to them)")
wt = s[:, 15]
if wt.sum() > 0:
    print(f"  block-steps in which a wave found a row missing: {100 * wt.sum() / ns.sum():.2f} %, {s[:, 6].sum() / wt.sum():.0f} ticks in the poll loop per such step (mean), "
          f"{s[:, 7].sum() / wt.sum():.1f} polls each; per wave min / max share {100 * (wt / ns).min():.1f} / {100 * (wt / ns).max():.1f} %")
    pw = s[:, 6] / np.maximum(wt, 1)
    print(f"  poll-loop ticks per wait, per wave: median {np.median(pw[wt > 0]):.0f}  95 % {np.percentile(pw[wt > 0], 95):.0f}  max {pw.max():.0f}")
e.close()
# per-wave speed without the waits: is somebody slower than the rest all the time?
work = (s[:, 8:14].sum(axis=1) - s[:, 6]) / ns
order = np.argsort(work)
print(f"  ticks per block-step without the waits, per wave: min {work.min():.0f}  5 % {np.percentile(work, 5):.0f}  median {np.median(work):.0f}  95 % {np.percentile(work, 95):.0f}  max {work.max():.0f}")
print("  slowest waves (workgroup index: ticks):", ", ".join(f"{int(i)}: {work[i]:.0f}" for i in order[-8:]))
print("  by workgroup index mod 8 (XCD), mean:", " ".join(f"{work[k::8].mean():.0f}" for k in range(8)))
print("  poll-loop ticks per wave by workgroup index mod 8, mean per step:", " ".join(f"{(s[k::8, 6] / ns[k::8]).mean():.0f}" for k in range(8)))
print("  by workgroup index, buckets of 64, mean:", " ".join(f"{work[k:k + 64].mean():.0f}" for k in range(0, len(work), 64)))
print("  workgroups slower than 1.1 x median:", int((work > 1.1 * np.median(work)).sum()), "of", len(work), "; indices", np.nonzero(work > 1.1 * np.median(work))[0][:40].tolist())
