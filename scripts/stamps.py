"""Where a split-layout launch spends its time: shader-clock stamps written by a diagnostic build
(-DDEMCZ_STAMPS, build_ab/stamps.so; never the shipped library).
usage: python scripts/stamps.py [K] [generations] [append_lag]   (run on the GPU box)"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
G = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lag = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, d = 1024, 5
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                   target=w["target"])
if lag:
    e.set_append_lag(lag)
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
import time
t0 = time.perf_counter()
e.run(1, G, 2.38)
e.synchronize()
wall_us = (time.perf_counter() - t0) * 1e6
lib = _lib.load()
ncons = (N + 7) // 8
nprod = ((N + 63) // 64) * 4 * (K * max(lag, 1))      # (rough: producer workgroups of the launch's successor)
nwg = ncons + nprod
buf = np.zeros((nwg, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), nwg)
assert rc == 0, rc
s = buf.astype(np.int64)
t00 = s[:, 0].min()
c = s[:ncons]
print(f"stamps of the last launch (K={K}, lag={lag}); shader-clock ticks; consumer workgroups: {ncons}, producer: {nprod}")
print(f"entry skew over all workgroups: {s[:,0].max() - t00} ticks; consumers enter at {np.mean(c[:,0]-t00):.0f} (mean) / {np.max(c[:,0]-t00)} (max)")
names = {3: "loads back (hand-off waits included), increments in LDS", 2: "next chunk's loads issued (launches that prefetch)",
         4: "generations of the chunk done", 7: "exit"}
print(f"  timed chunk starts at +{np.mean(c[:,5]-c[:,0]):.0f} (mean); chunk-relative below, exit launch-relative")
for i in (3, 2, 4, 7):
    dt = c[:, i] - (c[:, 0] if i == 7 else c[:, 5])
    print(f"  consumer +{np.mean(dt):8.0f} mean  {np.min(dt):6d} min {np.max(dt):6d} max   {names[i]}")
nch = c[:, 14].astype(float)
if nch.min() > 0:
    print("  sums over all chunks of the launch, ticks per chunk (mean over consumers / slowest consumer):")
    for j, nm in ((8, "chunk start -> rows there, increments in LDS (waits for unpublished rows included)"),
                  (9, "-> next chunk's loads issued"), (10, "-> generations done"), (11, "-> append + LDS hand-off")):
        v = c[:, j] / nch
        print(f"    {v.mean():8.0f} {v.max():8.0f}   {nm}")
    tot = (c[:, 8] + c[:, 9] + c[:, 10] + c[:, 11]) / nch
    print(f"    {tot.mean():8.0f} {tot.max():8.0f}   per chunk in all; chunks per launch {nch.mean():.0f}")
    print(f"    chunks that had to ask again for a row: {100 * (c[:, 12] / nch).mean():.1f} % (mean over consumers), polls per such chunk "
          f"{(c[:, 13].sum() / max(c[:, 12].sum(), 1)):.1f}")
p = s[ncons:nwg]
ok = p[:, 7] > 0
print(f"  producer: {ok.sum()} workgroups wrote; body {np.mean((p[ok,7]-p[ok,0])):.0f} mean {np.max(p[ok,7]-p[ok,0])} max; last exit at {np.max(p[ok,7])-t00} after the first entry")
print(f"  last consumer exit at {np.max(c[:,7]) - t00} after the first entry")
print(f"  wall time of the whole call ({G} generations, all launches): {wall_us:.0f} us; ticks of the last launch's consumers: {np.mean(c[:,7]-c[:,0]):.0f}")
e.close()
