"""Where a generation (lr16) / a step (lr8s: up to two generations) of the regression kernel goes: shader-clock sums per wave from a -DDEMCZ_STAMPS build.
usage: python scripts/lr_stamps.py [nobs] [N]"""
import ctypes as C, os, subprocess, sys
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
nobs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d, G = 10, 682
w = demc.workloads.linreg_problem(d, N, nobs=nobs)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=10, Mcap=M0 + N * (G // 10 + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
T = np.array([demc.tempbaseline(g, G, 3, 1e-3) for g in range(1, G + 1)])
e.run(1, G, w["gamma"], T); e.synchronize()
lib = _lib.load()
per_wg = 16 if os.environ.get('DEMCZ_NO_LR_SPEC') or (N + 7) // 8 > 256 else 8
nw = (N // per_wg) * 4
buf = np.zeros((nw, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), nw) == 0
s = buf.astype(np.float64)
ng = s[:, 14]
steps = s[:, 15]
if per_wg == 8:
    print(f"eight chains per workgroup: {steps.mean():.1f} steps (max {steps.max():.0f}) for {int(ng[0])} generations = {ng[0] / steps.mean():.3f} generations per step; below: ticks per STEP")
    ng = steps
names = ["proposal, next generation's loads issued, previous generation's history stored", "matrix instructions + partial to LDS", "workgroup barrier",
         "partials read, tree, accept, append", "wait for this generation's rows and records, sentinel poll", "between steps"]
print(f"nobs={nobs} N={N}: last launch {int(ng[0])} generations; ticks per generation, mean over waves / wave 0 of the workgroups / max")
for i in (4, 0, 1, 2, 3, 5):
    v = s[:, 8 + i] / ng
    print(f"  {v.mean():8.0f} {v[0::4].mean():8.0f} {v.max():8.0f}   {names[i]}")
if per_wg == 8:
    for i, nm in enumerate(["proposals", "second read of missing rows", "loads of g+2, g+3 issued", "both trees", "tests, state", "window moved"]):
        print(f"     sub {nm:32s} {(s[:, i] / ng).mean():8.0f}")
tot = s[:, 0:6].sum(axis=1) / ng + s[:, 8:14].sum(axis=1) / ng
print(f"  {tot.mean():8.0f} ticks per generation in all")
e.close()
