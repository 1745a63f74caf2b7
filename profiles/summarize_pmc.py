"""Aggregate rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv) into one small
per-kernel table.  Usage: python profiles/summarize_pmc.py <pmc_mfma_dir> <pmc_fetch_dir> <pmc_write_dir> > out.csv

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE
reports half of the bytes of wide (16 B/lane) streaming reads -> the `fetch_MB_x2` column doubles
it (upper bound for kernels with narrow reads).  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); effective clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import glob
import sys


def load(d):
    cc = glob.glob(d + "/*/*_counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = list(csv.DictReader(open(cc)))
    grids = collections.defaultdict(set)
    for r in rows:
        grids[r["Kernel_Name"]].add(r["Grid_Size"])
    for r in rows:
        k = r["Kernel_Name"]
        if len(grids[k]) > 1 and SPLIT_BY_GRID:      # same kernel at several sizes (one per layer): keep them apart
            k = k[:70] + " [grid " + r["Grid_Size"] + "]"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur:" + r["Dispatch_Id"]] = [dur.get(r["Dispatch_Id"], 0)]
    out = {}
    for k, v in agg.items():
        ds = [x[0] for n, x in v.items() if n.startswith("_dur:")]
        out[k] = {n: sum(x) / len(x) for n, x in v.items() if not n.startswith("_dur:")}
        out[k]["dur_ns"] = sum(ds) / max(len(ds), 1)
        out[k]["launches"] = len(ds)
    return out


SPLIT_BY_GRID = True
m, f, w = (load(a) for a in sys.argv[1:4])
wr = csv.writer(sys.stdout)
wr.writerow(["kernel", "launches", "avg_us", "eff_clock_GHz", "mfma_pipe_util", "fetch_MB_raw", "fetch_MB_x2", "write_MB",
             "hbm_GBps_x2"])
for k, v in sorted(m.items(), key=lambda kv: -kv[1]["dur_ns"] * kv[1]["launches"]):
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    fe = f.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / 1e6
    wb = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
    us = v["dur_ns"] / 1e3
    wr.writerow([k[:110] if "[grid" not in k else k, v["launches"], round(us, 1), round(gui / 8 / v["dur_ns"], 2) if v["dur_ns"] else 0,
                 round(mf / 1024 / (gui / 8), 3) if gui else 0, round(fe, 2), round(2 * fe, 2), round(wb, 2),
                 round((2 * fe + wb) / us * 1e3, 0) if us else 0])
