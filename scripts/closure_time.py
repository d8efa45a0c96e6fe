"""bench.py's `configs.closure` on its own: the host-closure mode at C2's shape, copies against pinned buffers.
usage: python scripts/closure_time.py [generations]"""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import demc_jl_amd as demc

r = bench.closure_row(demc, 31953150, 0, gens=int(sys.argv[1]) if len(sys.argv) > 1 else 300)
print(json.dumps(r, indent=1))
