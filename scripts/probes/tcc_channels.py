"""Per-instance (XCC x L2 channel) values of TCC counters for the window kernel, from rocprofv3's JSON output.
usage (GPU box): cd /tmp; rocprofv3 --pmc TCC_REQ TCC_TAG_STALL TCC_BUSY --kernel-trace --output-format json -d <out> -- python3 bench.py ...
                 python scripts/probes/tcc_channels.py <out> [kernel substring]"""
import json
import sys
from collections import defaultdict
from pathlib import Path

out, sub = Path(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "window_kernel")
files = sorted(out.rglob("*_results.json"))
assert files, f"no *_results.json under {out}"
doc = json.loads(files[0].read_text())
sdk = doc["rocprofiler-sdk-tool"][0]
if "--schema" in sys.argv:
    def walk(o, ind=0, key=""):
        if isinstance(o, dict):
            for k, v in list(o.items())[:40]:
                print(" " * ind + k + ": " + type(v).__name__ + (f"[{len(v)}]" if isinstance(v, (list, dict)) else f" = {str(v)[:60]}"))
                if ind < 8: walk(v, ind + 2, k)
        elif isinstance(o, list) and o:
            walk(o[0], ind + 2, key)
    for k in sdk:
        if k not in ("metadata", "agents"):
            print("==", k); walk(sdk[k], 2, k)
    sys.exit(0)
ksym = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in sdk["kernel_symbols"]}
inst = {}            # counter id -> (name, [(xcc, channel) of its instances, in the order the file lists them])
for c in sdk["counters"]:
    order = []
    for it in c.get("instances", []):
        dims = {dd["dimension_name"]: dd["index"] for dd in it["dimensions"]}
        order.append((dims.get("DIMENSION_XCC", 0), dims.get("DIMENSION_INSTANCE", 0)))
    inst[c["id"]["handle"]] = (c["name"], order)
acc = defaultdict(lambda: defaultdict(float))
nd = 0
for rec in sdk["callback_records"]["counter_collection"]:
    di = rec["dispatch_data"]["dispatch_info"]
    if sub not in ksym.get(di["kernel_id"], ""):
        continue
    nd += 1
    seen = defaultdict(int)
    for r in rec["records"]:          # (a record carries its counter and value only: the n-th record of a counter is taken to be its n-th instance)
        h = r["counter_id"]["handle"]
        name, order = inst.get(h, (str(h), []))
        key = order[seen[h]] if seen[h] < len(order) else (99, seen[h])
        seen[h] += 1
        acc[name][key] += r["value"]
print(f"{nd} dispatches of *{sub}*; per dispatch, rows = XCC, columns = the XCC's L2 channel")
import numpy as np
for name, d in sorted(acc.items()):
    keys = sorted(d)
    a, b = sorted(set(k[0] for k in keys)), sorted(set(k[1] for k in keys))
    M = np.array([[d.get((x, y), 0.0) / max(nd, 1) for y in b] for x in a])
    print(f"{name}: total {M.sum():.0f}")
    for x, row in zip(a, M):
        print(f"  {x:2d}: " + " ".join(f"{v:8.0f}" for v in row))
    print("  sum over rows:    " + " ".join(f"{v:8.0f}" for v in M.sum(axis=0)))
