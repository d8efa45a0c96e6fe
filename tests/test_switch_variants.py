"""The kernels' A/B switches (compile-time -D macros) select code that the shipped library does not contain: the forms measured
against each other in profiles/r04s_dpp.txt and r04o_history_store.txt.  One build with every switch flipped keeps that code
compiling (CPU tier) and bit-identical to the oracle (GPU tier: a few of the long parity tests against the flipped build, in one
child process -- the library is chosen at import time by DEMCZ_LIB, never a CPU path)."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
FLIPPED = ROOT / "build_ab" / "flipped.so"
# W / increments through scalar loads and LDS rows again in window_kernel_pw; the chain's hand-offs through LDS and the record's by
# DPP in window_kernel_mlb; candidate adds by lanes in window_kernel_ps2 / ps2d; the one-chain kernel's history through the ring;
# the sixteen-lane regression kernel's proposals as wave-uniform register copies instead of by lanes
SWITCHES = ["-DPW_WDPP=0", "-DPW_DDPP=0", "-DMLB_DPP_RECORD=1", "-DPS2_DDPP=1", "-DPS2_HRING_ONE=1", "-DML_LRDPP=0"]


def test_flipped_switches_build():
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "build_variant.py"), "flipped"] + SWITCHES, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert FLIPPED.exists()


@pytest.mark.gpu
def test_flipped_switches_are_bit_identical_to_the_oracle():
    if not FLIPPED.exists():
        pytest.skip("build_ab/flipped.so not built (the CPU tier builds it)")
    env = dict(os.environ, DEMCZ_LIB=str(FLIPPED))
    sel = ["tests/test_gpu_long_oracle.py::test_wave_per_chain_regular_launches_equal_oracle",
           "tests/test_gpu_long_oracle.py::test_c3_block_updates_long_run_equals_oracle",
           "tests/test_gpu_dual.py", "tests/test_gpu_live.py::test_forced_handoff_timeout_is_redone_bit_exact",
           "tests/test_gpu_parity.py::test_regression_target_any_dimension_sixteen_lanes"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=str(ROOT), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
