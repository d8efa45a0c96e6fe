"""Where a pass of the steady-state wave-per-chain kernel (demcz_kernels_ps2.h) spends its time: shader-clock sums per segment
written by a diagnostic build (-DDEMCZ_STAMPS, build_ab/stamps.so; never the shipped library).  A stamp drains the wave's
outstanding LDS operations, so the segments are proportions, not absolute costs.
usage: python scripts/ps2_stamps.py [N] [K] [generations] [M0]   (run on the GPU box; M0: rows of a synthetic initial archive)"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
so = ROOT / "build_ab" / "stamps.so"
os.environ["DEMCZ_LIB"] = str(so)          # (before anything imports demc_jl_amd: _lib reads it at import)
src = ROOT / "demc.jl_amd" / "csrc"
if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in src.glob("*")):
    so.parent.mkdir(exist_ok=True)
    sys.path.insert(0, str(ROOT))
    from demc_jl_amd import _lib as _build
    _build.build_lib(so, extra=["-DDEMCZ_STAMPS"])          # (every translation unit, in parallel: demc.jl_amd/_lib.py)
sys.path.insert(0, str(ROOT))
import numpy as np
import demc_jl_amd as demc
from demc_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
M0big = int(sys.argv[4]) if len(sys.argv) > 4 else 0
d = 5
w = demc.workloads.mvnormal_problem(d, N)
if M0big:
    w["Zinit"] = np.asfortranarray(w["mu"] + 0.1 * np.random.default_rng(0).standard_normal((M0big, d)))
M0 = w["Zinit"].shape[0]
e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)], eps_scale=w["eps_scale"], seed=1,
                   target=w["target"])
assert e.info()["lanes_per_chain"] == 164
e.set_state(w["Zinit"][-N:], None, w["Zinit"])
e.run(1, G // 2, 2.38)
e.set_kernel_timing(True)
e.run(G // 2 + 1, G, 2.38)
nl, ms = e.get_kernel_time()
lib = _lib.load()
kname = e.kernel_name()
NCHAINS = N
if "ps2d" in kname:          # two chains to a wave: the stamps are per WAVE
    N = (N + 1) // 2
print(f"kernel {kname}: {NCHAINS} chains, {N} chain waves")
buf = np.zeros((N, 16), dtype=np.uint64)
lib.demcz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = lib.demcz_debug_read_stamps(e._h, buf.ctypes.data_as(C.c_void_p), N)
assert rc == 0, rc
s = buf.astype(np.float64)
assert np.all(s[:, 15] == 2), "the last launch was not window_kernel_ps2's"
n = s[:, 14]
print(f"N={N} K={K} M0={M0}: last launch = {G - G // 2} generations in {ms * 1e3:.1f} us (stamped build), {n.mean():.0f} passes per chain wave")
print(f"  whole launch: {s[:, 8].mean():.0f} shader clocks mean, {s[:, 8].max():.0f} max = {(s[:, 8] / n).mean():.0f} per pass")
names = ["history store, DMA wait, raw values asked for, candidate adds", "log-density, the next pass's increments",
         "bpermute asked for, table write, DMA issue, next pass's rows asked for", "accept tests, path (waits for the bpermute)",
         "winner, state + history values back from the table", "boundary: row to the publisher",
         "waits for rows not yet published (round 5: timed only in the passes that wait -- until round 4 this line was the whole segment "
         "between two stamps, which every pass pays 100-200 clocks of stamp overhead for, and 'per waiting pass' divided THAT by the waiting passes)"]
for i, nm in enumerate(names):
    print(f"  {(s[:, i] / n).mean():8.0f} per pass  ({100 * s[:, i].sum() / s[:, 8].sum():5.1f} %)  {nm}")
print(f"  passes that waited for a row: {100 * (s[:, 11] / n).mean():.2f} %")
print(f"  polls of a full history ring (the publisher {'PS2_HSLOTS'} passes behind): {s[:, 7].sum() / n.sum():.3f} per pass")
if s[:, 11].sum() > 0:
    print(f"  per waiting pass: {s[:, 6].sum() / s[:, 11].sum():.0f} clocks in the wait segment, {s[:, 12].sum() / s[:, 11].sum():.1f} polls "
          f"(a poll = one round of past-the-caches loads of the rows still missing + s_sleep), {s[:, 6].sum() / max(s[:, 12].sum(), 1):.0f} clocks per poll")
STAMP_WGS = 65536
pb = np.zeros((STAMP_WGS // 2 + (N + 3) // 4, 16), dtype=np.uint64)
if lib.demcz_debug_read_stamps(e._h, pb.ctypes.data_as(C.c_void_p), pb.shape[0]) == 0:
    pb = pb[STAMP_WGS // 2:].astype(np.float64)
    if pb[:, 0].sum() > 0:
        it = pb[:, 0]
        print(f"  publisher waves: {it.mean():.0f} loop rounds a launch ({(pb[:, 4] / it).mean():.0f} clocks each), {pb[:, 3].mean():.0f} of them idle (s_sleep); "
              f"waiting for room in the store queue {100 * pb[:, 1].sum() / pb[:, 4].sum():.1f} % of its time, reading the ring + issuing a pass's history stores "
              f"{(pb[:, 2] / np.maximum(pb[:, 5], 1)).mean():.0f} clocks per pass ({100 * pb[:, 2].sum() / pb[:, 4].sum():.1f} % of its time); {pb[:, 5].mean():.0f} passes of history, {pb[:, 6].mean():.0f} boundary rows per lane")
if buf[:, 9].max() > 0:      # window_kernel_ps2: when each chain wave began and ended (100 MHz clock common to all CUs)
    t0, t1 = buf[:, 9].astype(np.int64), buf[:, 10].astype(np.int64)
    b = (t0 - t0.min()) / 100.0
    e = (t1 - t0.min()) / 100.0
    print(f"  chain waves begin over {b.max():.1f} us (median {np.median(b):.1f}, 90 % by {np.quantile(b, 0.9):.1f}); end between {e.min():.1f} and {e.max():.1f} us")
    cidx = np.arange(N)
    print("  mean end by wave of the workgroup (chain mod 4): " + " ".join(f"{e[cidx % 4 == w].mean():.1f}" for w in range(4)))
    print("  mean end by XCD (chain // (N/8)):               " + " ".join(f"{e[cidx // (N // 8) == x].mean():.1f}" for x in range(8)))
    wsum = s[:, 6] / 2100.0      # clocks -> us at ~2.1 GHz
    print(f"  own waits per chain: mean {wsum.mean():.1f} us, min {wsum.min():.1f}, max {wsum.max():.1f}; corr(end, waits) = {np.corrcoef(e, wsum)[0, 1]:.2f}")
    work = (s[:, 8] - s[:, 6]) / 2100.0
    print(f"  launch minus own waits per chain: mean {work.mean():.1f} us, min {work.min():.1f}, max {work.max():.1f}; by wave: " + " ".join(f"{work[cidx % 4 == w].mean():.1f}" for w in range(4)))
    wg = work.reshape(-1, 4)              # chains of a workgroup (one CU) are consecutive
    print(f"  work: std over all chains {work.std():.2f} us; std of workgroup means {wg.mean(axis=1).std():.2f}; mean std inside a workgroup {wg.std(axis=1).mean():.2f}")
    order = np.argsort(wg.mean(axis=1))
    print("  slowest workgroups (index: mean work us): " + ", ".join(f"{i}: {wg[i].mean():.1f}" for i in order[-6:]) + "; fastest: " + ", ".join(f"{i}: {wg[i].mean():.1f}" for i in order[:4]))
    q = np.quantile(wg.mean(axis=1), [0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
    print("  workgroup mean work, quantiles 0/10/25/50/75/90/100 %: " + " ".join(f"{v:.1f}" for v in q))
    nwg = wg.shape[0]
    print("  workgroup mean work by workgroup index mod 8: " + " ".join(f"{wg.mean(axis=1)[np.arange(nwg) % 8 == x].mean():.1f}" for x in range(8)))
    print("  workgroup mean work by eighths of the index range: " + " ".join(f"{wg.mean(axis=1)[np.arange(nwg) * 8 // nwg == x].mean():.1f}" for x in range(8)))
    if buf[:, 13].max() > 0:     # where every chain wave ran
        hw = buf[:, 13]
        simd, cu, sh, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
        place = (xcc * 8 + se) * 32 + sh * 16 + cu            # one number per CU
        wgp = place.reshape(-1, 4)
        same = np.all(wgp == wgp[:, :1], axis=1)
        uniq, cnt = np.unique(wgp[:, 0], return_counts=True)
        shared = np.isin(wgp[:, 0], uniq[cnt > 1])
        wm = wg.mean(axis=1)
        print(f"  placement: {len(uniq)} distinct CUs hold the {nwg} workgroups ({int(same.sum())} with all four chain waves on one CU); "
              f"{int(shared.sum())} workgroups share their CU with another: mean work {wm[shared].mean() if shared.any() else float('nan'):.1f} us against {wm[~shared].mean():.1f} alone")
        print("  workgroups per XCC id: " + " ".join(f"{int((xcc.reshape(-1, 4)[:, 0] == x).sum())}" for x in range(8)) +
              "; distinct CUs used per XCC: " + " ".join(f"{len(np.unique(wgp[:, 0][xcc.reshape(-1, 4)[:, 0] == x]))}" for x in range(8)))
        sm = simd.reshape(-1, 4)
        print(f"  workgroups whose four chain waves sit on four different SIMDs: {int((np.sort(sm, axis=1) == np.arange(4)).all(axis=1).sum())} of {nwg}")
        slow = wm > np.median(wm) + 4 * np.median(np.abs(wm - np.median(wm))) + 5.0
        if slow.any():
            xw = xcc.reshape(-1, 4)[:, 0]
            print(f"  {int(slow.sum())} workgroups are slow (work > median + 4 MAD + 5 us): XCC ids " + " ".join(f"{x}:{int((xw[slow] == x).sum())}" for x in range(8)) +
                  "; SE ids " + " ".join(f"{x}:{int((se.reshape(-1, 4)[:, 0][slow] == x).sum())}" for x in range(8)))
            seg = (s[:, :7] / n[:, None]).reshape(-1, 4, 7).mean(axis=1)
            print("  clocks per pass by segment, slow workgroups:  " + " ".join(f"{v:6.0f}" for v in seg[slow].mean(axis=0)))
            print("  clocks per pass by segment, the others:       " + " ".join(f"{v:6.0f}" for v in seg[~slow].mean(axis=0)))
            ids = sorted((int(xw[i]), int(se.reshape(-1, 4)[i, 0]), int(sh.reshape(-1, 4)[i, 0]), int(cu.reshape(-1, 4)[i, 0]), i) for i in np.nonzero(slow)[0])
            print("  slow workgroups as xcc.se.sh.cu(workgroup index): " + " ".join(f"{a}.{b}.{c}.{d}({i})" for a, b, c, d, i in ids))
            allids = sorted(set((int(b), int(c), int(d)) for b, c, d in zip(se.reshape(-1, 4)[:, 0][xw == 0], sh.reshape(-1, 4)[:, 0][xw == 0], cu.reshape(-1, 4)[:, 0][xw == 0])))
            print("  XCC 0's CUs as se.sh.cu: " + " ".join(f"{b}.{c}.{d}" for b, c, d in allids))
            li = np.nonzero(slow)[0]
            print("  slow workgroups' index mod 32: " + " ".join(f"{x}:{int((li % 32 == x).sum())}" for x in range(32) if (li % 32 == x).any()))
