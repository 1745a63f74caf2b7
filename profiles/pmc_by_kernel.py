import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {n: round(sum(x)/len(x)) for n, x in v.items()})
