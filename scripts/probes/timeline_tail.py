import csv, glob, sys
f = glob.glob("gpurun_out/tl_trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels of the first config with comm (find ncclDevKernel / append_batch)
idx = [i for i, r in enumerate(rows) if "append_batch" in r["Kernel_Name"]]
if not idx: print("no append_batch"); sys.exit()
a = idx[len(idx)//6]   # somewhere inside the E=10 comm run
t0 = int(rows[a-12]["Start_Timestamp"])
for r in rows[a-12:a+14]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} us  q={r.get('Queue_Id','?')}  {r['Kernel_Name'][:70]}")
