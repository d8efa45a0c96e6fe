#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_{trace,fetch,write} (scripts/collect_profiles.sh) into profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  <tag>_traffic.json       per-launch FETCH_SIZE / WRITE_SIZE of every kernel, raw and corrected
Correction (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a coalesced streaming read, WRITE_SIZE is exact.  Calibration in THIS access
pattern: rhat_moments_kernel streams a known byte count (N*d*w*8, 8 B per lane, coalesced) and
its FETCH_SIZE reads exactly half of it, so streaming kernels get x2; the window kernel's reads
are single-line random gathers (one 64-byte request per row) for which the raw count matches the
lines touched, so it is reported raw with the x2 figure beside it as an upper bound."""
import csv, glob, json, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
stats = glob.glob(f"gpurun_out/{tag}_trace/*/*kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/{tag}_{name}/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[ctr + "_KiB_per_launch_mean"] = sum(v) / len(v)
        out[k][ctr + "_KiB_per_launch_last"] = v[-1]
        out[k]["launches"] = len(v)
for k, v in out.items():
    f_, w_ = v.get("FETCH_SIZE_KiB_per_launch_mean", 0.0), v.get("WRITE_SIZE_KiB_per_launch_mean", 0.0)
    v["bytes_per_launch_raw"] = (f_ + w_) * 1024
    v["bytes_per_launch_fetch_x2"] = (2 * f_ + w_) * 1024
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1, sort_keys=True)
json.dump({"tag": tag, "workload": "bench.py defaults (C2: N=1024, d=5, K=10)", "kernels": out},
          open("profiles/latest_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "window" in k}, indent=1))
