#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel, from the disassembly of a build (no GPU needed):
    python scripts/isa_loop.py <lib.so> <mangled-name substring> [--dump]
Finds every backward branch in the kernel, takes the LONGEST loop body (the pass loop of the window kernels), and
counts its instructions by unit: SALU / VALU / LDS / VMEM / SMEM / branch / waitcnt / other."""
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def disassemble(lib):
    """The library is linked from several translation units (round 5): its .hip_fatbin section holds one offload bundle per unit,
    each with a gfx950 code object -- all of them are disassembled."""
    out = []
    with tempfile.TemporaryDirectory() as td:
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
                out.append(subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout)
    return "\n".join(out)


def kernel_insts(text, needle):
    out, on = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if on:
                break
            on = needle in m.group(1)
            continue
        if on:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m:
                out.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def unit(op):
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")): return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")): return "SMEM"
    if op.startswith(("s_nop", "s_sleep", "s_setprio", "s_barrier")): return "other"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_"): return "VALU"
    return "other"


def loops(insts):
    addr_ix = {a: i for i, (a, _, _) in enumerate(insts)}
    res = []
    for i, (a, op, args) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            try:
                off = int(args.split()[0])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_ix:
                res.append((addr_ix[tgt], i))
    return res


if __name__ == "__main__":
    lib, needle = sys.argv[1], sys.argv[2]
    insts = kernel_insts(disassemble(lib), needle)
    ls = sorted(loops(insts), key=lambda t: t[0] - t[1])
    print(f"{len(insts)} instructions in the kernel; loops (start, end, length): {[(a, b, b - a + 1) for a, b in ls[:6]]}")
    if not ls:
        sys.exit(0)
    a, b = ls[0]
    body = insts[a:b + 1]
    c = Counter(unit(op) for _, op, _ in body)
    print("longest loop:", dict(c), "total", len(body))
    spill = sum(1 for _, op, _ in body if op in ("v_writelane_b32", "v_readlane_b32"))
    print("v_readlane/v_writelane in it (SGPR spills + the state's own readlanes):", spill, "; scratch ops:",
          sum(1 for _, op, _ in body if op.startswith("scratch_")))
    if "--dump" in sys.argv:
        for ad, op, args in body:
            print(f"{ad:08x}  {op:32s} {args}")
