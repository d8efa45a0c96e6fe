"""Build the library with extra -D switches into build_ab/<name>.so -- the other side of an A/B (never the shipped library).
usage: python scripts/build_variant.py <name> [-DSWITCH ...]   then   DEMCZ_LIB=build_ab/<name>.so python bench.py ..."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from demc_jl_amd import _lib

name, extra = sys.argv[1], sys.argv[2:]
out = ROOT / "build_ab" / f"{name}.so"
out.parent.mkdir(exist_ok=True)
_lib.build_lib(out, extra=extra)
print(out)
