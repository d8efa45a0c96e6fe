"""Import alias for the package directory ``demc.jl_amd/`` (a dotted directory name cannot be
imported by name): ``import demc_jl_amd`` loads that package under this module's name."""
import importlib.util as _u
import sys as _sys
from pathlib import Path as _P

_dir = _P(__file__).resolve().parent / "demc.jl_amd"
_spec = _u.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
