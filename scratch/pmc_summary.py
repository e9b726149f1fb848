import csv, glob, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "nisshi" not in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                print(f"{k:45s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
