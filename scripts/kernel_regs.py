#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in a build of the HIP library, from the code object's own metadata
(no GPU needed):  python scripts/kernel_regs.py [lib.so] [name filter]
Columns: VGPRs, AGPRs, SGPRs, SGPR spills, VGPR spills, scratch bytes, static LDS bytes."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_table(lib):
    notes = ""
    with tempfile.TemporaryDirectory() as td:      # (one offload bundle per translation unit of the library: all of them)
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        blob = fat.read_bytes()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
        for n, a in enumerate(starts):
            part, co = Path(td) / f"b{n}.bin", Path(td) / f"b{n}.co"
            part.write_bytes(blob[a:starts[n + 1] if n + 1 < len(starts) else len(blob)])
            r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={part}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
            if r.returncode == 0 and co.exists() and co.stat().st_size > 0:
                notes += subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        def f(key, default="0"):
            m = re.search(rf"\.{key}:\s*(\S+)", blk)
            return m.group(1) if m else default
        name = f("name", "?")
        try:
            name = subprocess.run([str(LLVM / "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        rows.append((name, int(f("vgpr_count")), int(f("agpr_count")), int(f("sgpr_count")), int(f("sgpr_spill_count")),
                     int(f("vgpr_spill_count")), int(f("private_segment_fixed_size")), int(f("group_segment_fixed_size"))))
    return rows


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else str(Path(__file__).resolve().parent.parent / "demc.jl_amd" / "libdemcz_hip.so")
    filt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'sspill':>6} {'vspill':>6} {'scratch':>7} {'LDS':>7}  kernel")
    for r in sorted(kernel_table(lib)):
        if filt in r[0]:
            print(f"{r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:6d} {r[5]:6d} {r[6]:7d} {r[7]:7d}  {r[0][:150]}")
