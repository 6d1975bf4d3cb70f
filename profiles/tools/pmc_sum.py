import sys, glob, csv, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        for k in acc:
            if "cbf" in k or "lowlevel" in k or "k_step_geometric" in k:
                print(k, {c: round(acc[k][c] / cnt[k][c]) for c in acc[k]}, "launches", max(cnt[k].values()))
