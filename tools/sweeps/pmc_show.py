import csv,glob,collections,sys
d, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"{d}/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f"{k:28s} {sum(v)/len(v):14.0f}  n={len(v)}")
