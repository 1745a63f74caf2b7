"""Print per-step kernel totals from a rocprofv3 --stats kernel_stats.csv (steps = arg 2)."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"{d}: total {tot:.1f} us/step, launches/step {sum(int(r['Calls']) for r in rows)/steps:.1f}")
for r in rows:
    if pat and not any(p in r["Name"] for p in pat.split(",")):
        continue
    print(f"  {r['Name'][:90]:90s} calls/step={int(r['Calls'])/steps:5.1f} avg={float(r['AverageNs'])/1e3:8.1f}us  per-step={float(r['TotalDurationNs'])/steps/1e3:8.1f}")
