import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "decode_fused" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for d in sorted(by):
        print(d, {k: f"{v:.4g}" for k, v in sorted(by[d].items())})
