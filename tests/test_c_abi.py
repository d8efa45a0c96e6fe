"""The drop-in boundary from plain C: tests/c_abi/abi_smoke.c includes include/demcz.h as C99 (-pedantic -Werror: the header
must be valid C, not just valid C++), links libdemcz_hip.so and runs create -> set_state -> run -> get_history -> get_state ->
destroy without Python in between.  CPU tier: it compiles, links and -- there being no GPU -- fails loudly with
DEMCZ_ERR_NO_DEVICE (exit 77: no CPU fallback).  GPU tier: it runs and its own checks pass."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _build(tmp_path):
    import demc_jl_amd as demc
    lib = Path(demc.LIB_PATH)
    assert lib.exists(), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_abi" / "abi_smoke.c"),
           "-o", str(exe), f"-L{lib.parent}", f"-l:{lib.name}", f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True)
    return exe


def test_header_is_c99_and_client_links(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is the gpu-marked test")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no" in r.stderr.lower(), (r.returncode, r.stderr)      # DEMCZ_ERR_NO_DEVICE: there is no CPU path


@pytest.mark.gpu
def test_c_client_runs_the_hot_path(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke OK" in r.stdout
