"""Run the C2 workload for G generations in one call and report the library's error text, if any.
usage: python scripts/live_diag.py [G] [K]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import demc_jl_amd as demc
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, d = 1024, 5
w = demc.workloads.mvnormal_problem(d, N)
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1, target=w["target"])
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
t0 = time.perf_counter()
try:
    e.run(1, G, 2.38); e.synchronize()
    print(f"ok: {G} generations, K={K}, {time.perf_counter()-t0:.3f} s, M={e.M}", flush=True)
except demc.DemczError as ex:
    print(f"FAILED after {time.perf_counter()-t0:.3f} s: {ex}", flush=True)
    import ctypes as C, os
    from demc_jl_amd import _lib
    lib = _lib.load()
    if hasattr(lib, "demcz_debug_read_stamps"):       # diagnostic build (DEMCZ_LIB=build_ab/stamps.so): progress of every consumer
        buf = np.zeros((N // 8, 8), dtype=np.uint64)
        lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), N // 8)
        prog = buf[:, 6].astype(np.int64) - 1000000
        vals, cnt = np.unique(prog, return_counts=True)
        print("consumer workgroups by the chunk (first generation, 0-based in the launch) they reached:", dict(zip(vals.tolist(), cnt.tolist())))
        behind = np.argsort(prog)[:8]
        print("workgroups furthest behind:", behind.tolist())
        for wg in behind:
            row = int(buf[wg, 4]) - 2000000000; uk = int(buf[wg, 3]) - 3000000000; sp = int(buf[wg, 2]) - 4000000000
            print(f"  workgroup {wg}: chunk {prog[wg]}, waiting for row {row} (u*100+p = {uk}), polls {sp}")
