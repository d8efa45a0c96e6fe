"""Mean of every PMC counter per kernel from a rocprofv3 counter_collection.csv: python scripts/pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "window"
agg = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):16.1f}  (n={len(agg[k])})")
