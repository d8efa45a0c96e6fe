"""Fine dimension sweep of the wave-per-chain consumer at N = 1024, K = 10 (diagnosis; bench.py's d_sweep is the reported one).
usage: python scripts/d_sweep_fine.py <target: mvn|iso> <d_from> <d_to> [generations]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import demc_jl_amd as demc

kind, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
for d in range(a, b + 1):
    w = demc.workloads.mvnormal_problem(d, 1024) if kind == "mvn" else demc.workloads.iso_quad_problem(d, 1024)
    r = bench.config_row(demc, f"{kind} d={d}", w, 1024, d, 10, [range(d)], 31953150, 0, gens=gens, anneal=(kind == "iso"))
    print(f"{r['workload']:12s} {r['kernel'].split('::')[1]:58s} {r['us_per_K_window_kernels']:6.2f} us/K-window  {r['value']:.3e} upd/s  "
          f"{100 * r['roofline']['frac']:5.2f} % HBM  live {r['live_launches']} redos {r['live_redos']} launches {r['launches']}", flush=True)
