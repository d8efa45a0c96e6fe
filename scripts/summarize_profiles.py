#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_{trace,fetch,write} (scripts/collect_profiles.sh) into profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  <tag>_traffic.json       per-launch FETCH_SIZE / WRITE_SIZE of every kernel, raw and corrected, plus -- for the
                           bench -- the shape of the profiled launches (chains, d, K, layout, generations per launch)
                           that bench.py compares its own launches with before it reports `roofline.traffic`
Correction (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a coalesced streaming read, WRITE_SIZE is exact.  Calibration of the window
kernel's OWN two read patterns (round 4: scripts/probes/fetch_calibration.hip through
scripts/collect_calibration.sh <tag>): random 64-byte-row gathers of three 16-byte pieces are counted
in full as lines (factor ~1.00 of rows x 64 B); the draw-record pieces (16 bytes a lane, consecutive
lanes consecutive addresses, every byte of a (field, chain) row once) are counted at ~0.51 of their
bytes -- like the plain stream (0.50).  The consumer's FETCH_SIZE is the sum of the two, and the record
bytes of a launch are known exactly ((d + 2) x chains x generations x 8), so
    fetched = (FETCH_SIZE - f_records x record_bytes) / f_gather + record_bytes
and `bytes_per_launch_calibrated` = that + WRITE_SIZE (+ the producer kernel's WRITE_SIZE).
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
# calibration factors: FETCH_SIZE of the probe's kernels / their known bytes (this tag's passes, else the last ones kept)
cal = None
try:
    known = None
    for ln in open(f"gpurun_out/{tag}_cal.json"):
        if ln.startswith("{"):
            known = json.loads(ln)
    fc = glob.glob(f"gpurun_out/{tag}_cal/*/*counter_collection.csv")
    if known and fc:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fc[0])):
            if r["Counter_Name"] == "FETCH_SIZE":
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
        mean = {k: sum(v) / len(v) for k, v in agg.items()}
        cal = {"f_gather_per_line_byte": mean["cal_gather"] / known["cal_gather"]["lines_bytes"],
               "f_records_per_unique_byte": mean["cal_records"] / known["cal_records"]["unique_bytes"],
               "f_stream": mean["cal_stream"] / known["cal_stream"]["unique_bytes"],
               "probe": "scripts/probes/fetch_calibration.hip", "known_bytes": known,
               "FETCH_SIZE_bytes": {k: mean[k] for k in ("cal_gather", "cal_records", "cal_stream")}}
        json.dump(cal, open("profiles/fetch_calibration.json", "w"), indent=1, sort_keys=True)
except FileNotFoundError:
    pass
if cal is None:
    try:
        cal = json.load(open("profiles/fetch_calibration.json"))
    except FileNotFoundError:
        cal = None
doc["calibration"] = cal
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
    if cal:
        # one calibrated number per launch of the window kernel (+ its producer beside it)
        rec_bytes = (shape["dim"] + 2) * shape["chains"] * shape["generations_per_launch"] * 8.0
        tot = 0.0
        parts = {}
        for pre in shape["kernel_prefixes"]:
            for k, v in out.items():
                if k.startswith(pre):
                    f_, w_ = v.get("FETCH_SIZE_KiB_per_launch_mean", 0.0) * 1024.0, v.get("WRITE_SIZE_KiB_per_launch_mean", 0.0) * 1024.0
                    if "window_kernel" in k:
                        fetched = (f_ - cal["f_records_per_unique_byte"] * rec_bytes) / cal["f_gather_per_line_byte"] + rec_bytes
                        parts[k] = {"fetch_calibrated": fetched, "of_which_records": rec_bytes, "write": w_}
                    else:
                        fetched = f_ / max(cal["f_stream"], 1e-9)
                        parts[k] = {"fetch_calibrated": fetched, "write": w_}
                    v["bytes_per_launch_calibrated"] = fetched + w_
                    tot += fetched + w_
                    break
        shape["bytes_per_launch_calibrated"] = tot
        shape["calibrated_parts"] = parts
    doc["launch_shape"] = shape
    # the warm-up call's first launch of the LIVE instantiation and the timed ones have the same shape; PMC means are over all of them
json.dump(doc, open(f"profiles/{tag}_traffic.json", "w"), indent=1, sort_keys=True)
if latest:
    json.dump(doc, open("profiles/latest_traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "window" in k}, indent=1))
print("launch_shape:", shape)
