"""Ratio table from the three SQ counter passes of profiles/experiments/pmc_sq.sh:  python profiles/sq_table.py gpurun_out/<tag> [kernel-name substring ...]
Per kernel (mean over launches): share of wave time waiting, pipe activity relative to SQ_BUSY_CYCLES, bank-conflict share of LDS time, LDS / VALU instructions per MFMA."""
import collections
import csv
import glob
import sys

out, pats = sys.argv[1], sys.argv[2:] or [""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"{out}/{p}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(q in r["Kernel_Name"] for q in pats):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# wave-cycle counters are quad-cycles summed over waves: ratios only.  wait_* / WAVE_CYCLES; *_active / BUSY_CYCLES; bank_conflict / LDS_IDX_ACTIVE; instructions per MFMA instruction")
for k in sorted(agg):
    v = {n: sum(x) / len(x) for n, x in agg[k].items()}
    g = lambda n: v.get(n, 0.0)  # noqa: E731
    wc, bc, mf = max(g("SQ_WAVE_CYCLES"), 1.0), max(g("SQ_BUSY_CYCLES"), 1.0), max(g("SQ_INSTS_MFMA"), 1.0)
    print(f"{k[:92]:92s} wait_any {g('SQ_WAIT_ANY') / wc:.2f} wait_inst_any {g('SQ_WAIT_INST_ANY') / wc:.2f} wait_inst_lds {g('SQ_WAIT_INST_LDS') / wc:.2f} | "
          f"lds_active/busy {g('SQ_ACTIVE_INST_LDS') / bc:.2f} valu_active/busy {g('SQ_ACTIVE_INST_VALU') / bc:.2f} mfma_busy/busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / bc:.2f} | "
          f"bank_conflict/lds_active {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1.0):.3f} | lds/mfma {g('SQ_INSTS_LDS') / mf:.2f} valu/mfma {g('SQ_INSTS_VALU') / mf:.2f} vmem_rd/mfma {g('SQ_INSTS_VMEM_RD') / mf:.3f}")
