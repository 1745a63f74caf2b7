"""Print a rocprofv3 --stats kernel summary (csv) sorted by total time:  python profiles/kstats.py <dir> [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{len(rows)} kernels, total {tot / 1e3:.1f} us")
for r in rows[:n]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.2f} us {float(r['Percentage']):6.2f}%")
