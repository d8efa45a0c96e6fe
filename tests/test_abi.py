"""The C-ABI shared library loads and exports exactly what include/demcz.h declares (no compute
calls: this runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "demcz.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(demcz_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(demc):
    demc.build()
    lib = ctypes.CDLL(str(demc.LIB_PATH))
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/demcz.h but not exported"
    assert sorted(demc.SYMBOLS) == declared, "demc.jl_amd/_lib.py SYMBOLS out of sync with the header"
    lib.demcz_abi_version.restype = ctypes.c_int32
    assert lib.demcz_abi_version() == 1


def test_config_struct_layout_matches_header(demc):
    from demc_jl_amd._lib import Config
    # LP64 layout of demcz_config: 2*8 + 2*4 + 2*8 + (4 + pad4) + 3*8 + 8 + 2*4 + 2*8 + 8 + 2*8 + 8 + 8 + 2*4
    assert ctypes.sizeof(Config) == 152
    assert Config.Mcap.offset == 24 and Config.seed.offset == 72 and Config.c0.offset == 104 and Config.stream.offset == 136


def test_no_silent_cpu_fallback(demc):
    """Without a HIP device demcz_create must fail loudly (DEMCZ_ERR_NO_DEVICE)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    w = demc.workloads.mvnormal_problem(5, 8)
    with pytest.raises(demc.DemczError) as ei:
        demc.demcz_sample(w["target"], w["Zinit"], 8, 10, 20, verbose=False)
    assert ei.value.code == 5 and "no CPU fallback" in str(ei.value)


def test_product_does_not_reference_oracle():
    """Nothing under demc.jl_amd/ may import, link or execute anything under oracle/."""
    for p in (ROOT / "demc.jl_amd").rglob("*"):
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".cpp"):
            txt = p.read_text()
            assert "oracle_py" not in txt and "libdemcz_oracle" not in txt and "demcz_oracle" not in txt, p


def test_two_chain_kernel_is_what_its_generator_makes():
    """demcz_kernels_ps2d.h is derived from demcz_kernels_ps2.h by scripts/gen_ps2d.py; a change to the one-chain kernel must be
    carried over (re-run the script) or the two drift apart."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_ps2d.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "demcz_kernels_ps2d.h differs from what scripts/gen_ps2d.py generates: " + r.stderr[-500:]
    # the whitening chains of window_kernel_pw (W by lanes, DPP row_newbcast) are generated text too
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_pw_wdpp.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "demcz_pw_wdpp_*.inc differ from what scripts/gen_pw_wdpp.py generates: " + r.stderr[-500:]
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_mlb_dpp.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "demcz_mlb_dpp_*.inc differ from what scripts/gen_mlb_dpp.py generates: " + r.stderr[-500:]
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "gen_ml_lrdpp.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "demcz_ml_lrdpp_*.inc differ from what scripts/gen_ml_lrdpp.py generates: " + r.stderr[-500:]
