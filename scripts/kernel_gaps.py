"""Gaps between consecutive window-kernel launches in a rocprofv3 --kernel-trace of the bench, and what ran in them.
usage: python scripts/kernel_gaps.py <dir with *_kernel_trace.csv> [kernel substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "window_kernel"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
win = [r for r in rows if sub in r[2]]
print(f"{len(win)} launches of *{sub}*; duration mean {sum(e - s for s, e, _ in win) / len(win) / 1e3:.1f} us")
gaps = []
for (s0, e0, _), (s1, e1, _) in zip(win, win[1:]):
    inside = [(s, e, n) for s, e, n in rows if s < s1 and e > e0 and sub not in n]
    gaps.append((s1 - e0, inside))
gs = sorted(g for g, _ in gaps)
print(f"gap end -> next start: median {gs[len(gs) // 2] / 1e3:.1f} us, mean {sum(gs) / len(gs) / 1e3:.1f} us, min {gs[0] / 1e3:.1f}, max {gs[-1] / 1e3:.1f}")
for g, inside in gaps[-6:]:
    print(f"  gap {g / 1e3:6.1f} us; kernels overlapping it: " + ", ".join(f"{n.split('(')[0].split('::')[-1][:28]} [{(e - s) / 1e3:.1f} us]" for s, e, n in inside))
