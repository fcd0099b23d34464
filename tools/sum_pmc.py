import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "conv_mfma" in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
