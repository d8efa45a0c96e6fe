"""Condense scripts/collect_rec_halves.sh's runs (gpurun_out/<tag>_rec<MiB>_{trace,fetch}) into one table: per record-buffer size,
the bench line's rate, the window kernel's and the producer's time per 1000 generations, and FETCH_SIZE (= TCC_EA0_RDREQ x 64 B) per
1000 generations.      usage: python scripts/rec_halves_summary.py [tag] > profiles/<tag>_rec_halves.txt"""
import csv, glob, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
WIN, PRO = "window_kernel_ps", "produce_kernel<5>"          # (ps2 for regular launches, the general ps for the others)


def bench_line(log):
    for ln in open(log, errors="replace"):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    return None


def dispatches(d, want_counter=False):
    out = {WIN: [], PRO: []}
    names = set()
    if want_counter:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for k in out:
                    if k in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                        out[k].append(float(r["Counter_Value"]) * 1024.0)      # (KiB)
    else:
        for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for k in out:
                    if k in r["Kernel_Name"]:
                        if k == WIN:
                            names.add(r["Kernel_Name"].split("(")[0].replace("void demcz::", ""))
                        out[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out['names'] = names
    return out


print(f"C2 (1024 chains, d = 5, K = 10), bench.py --no-sweep --no-configs --steps 20 --warmup 5: the proposal records produced in pieces of")
print("DEMCZ_REC_MIB MiB (a launch covers the generations one record buffer holds; two buffers: a launch's producer fills one while the")
print("window kernel beside it reads the other).  Times from rocprofv3 --kernel-trace, FETCH_SIZE from its own --pmc pass; sums over ALL")
print("dispatches of the command divided by its generations (every launch of the command belongs to the 25 slabs + the tuning run).")
print()
print(f"{'MiB':>4} {'gens/launch':>11} {'launches':>8} {'updates/s':>10} {'median':>10} {'ms/step':>8} {'window us/1000 gens':>20} {'producer us/1000':>17} {'window FETCH MB/1000 gens':>26} {'producer FETCH MB/1000':>23}")
for mib in (64, 32, 16, 8):
    tr, fe = ROOT / "gpurun_out" / f"{tag}_rec{mib}_trace", ROOT / "gpurun_out" / f"{tag}_rec{mib}_fetch"
    if not tr.exists():
        continue
    line = bench_line(str(tr) + ".log")
    t, c = dispatches(tr), dispatches(fe, True)
    steps = (line["steps"] + line["warmup"]) if line else 25
    gens = 1000.0 * steps
    n = len(t[WIN])
    print(f"{mib:>4} {int(round(gens / max(1, n))):>11} {n:>8} {line['value'] if line else 0:>10.3g} {line.get('value_median', 0) if line else 0:>10.3g} {line['ms_per_step'] if line else 0:>8.4f} "
          f"{sum(t[WIN]) / gens * 1000:>20.1f} {sum(t[PRO]) / gens * 1000:>17.1f} {sum(c[WIN]) / gens * 1000 / 1e6:>26.1f} {sum(c[PRO]) / gens * 1000 / 1e6:>23.2f}   " + ", ".join(sorted(t["names"])))
