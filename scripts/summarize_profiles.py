#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_{trace,fetch,write} (scripts/collect_profiles.sh) into profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  <tag>_traffic.json       per-launch FETCH_SIZE / WRITE_SIZE of every kernel, raw and corrected, plus -- for the
                           bench -- the shape of the profiled launches (chains, d, K, layout, generations per launch)
                           that bench.py compares its own launches with before it reports `roofline.traffic`
Correction (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a coalesced streaming read, WRITE_SIZE is exact.  Calibration in THIS access
pattern: rhat_moments_kernel streams a known byte count (N*d*w*8, 8 B per lane, coalesced) and
its FETCH_SIZE reads exactly half of it, so streaming kernels get x2; the window kernel's reads
are single-line random gathers (one 64-byte request per row) for which the raw count matches the
lines touched, so it is reported raw with the x2 figure beside it as an upper bound.
usage: summarize_profiles.py <tag> [latest]      (`latest`: also write profiles/latest_traffic.json, the file bench.py reads)"""
import csv, glob, json, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
latest = len(sys.argv) > 2 and sys.argv[2] == "latest"
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
doc = {"tag": tag, "kernels": out}
# the bench's own JSON line (under the trace pass) tells what the profiled launches looked like
shape = None
try:
    for ln in open(f"gpurun_out/{tag}_trace.log"):
        if ln.startswith("{") and '"roofline"' in ln:
            b = json.loads(ln)
            c, r = b["config"], b["roofline"]
            shape = {"chains": c["chains_total"] // b["n_gpus"], "dim": c["dim"], "K": c["K"], "lanes_per_chain": c["lanes_per_chain"],
                     "generations_per_launch": r["generations_per_launch"],
                     # the kernels one window launch consists of: the wave-per-chain layout runs its producer half as a kernel of its own
                     "kernel_prefixes": (["void demcz::window_kernel_ps2<0, %d, true, false>" % c["dim"], "void demcz::produce_kernel<%d>" % c["dim"]]
                                         if c["lanes_per_chain"] == 164 else ["void demcz::window_kernel_pc8<0, %d, true, false>" % c["dim"]]),
                     "avg_launch_us_under_profiler": r["avg_launch_us"], "value_under_profiler": b["value"]}
            doc["workload"] = c["workload"]
except FileNotFoundError:
    pass
if shape:
    doc["launch_shape"] = shape
    # the warm-up call's first launch of the LIVE instantiation and the timed ones have the same shape; PMC means are over all of them
json.dump(doc, open(f"profiles/{tag}_traffic.json", "w"), indent=1, sort_keys=True)
if latest:
    json.dump(doc, open("profiles/latest_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "window" in k}, indent=1))
print("launch_shape:", shape)
