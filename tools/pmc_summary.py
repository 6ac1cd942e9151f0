"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output directories."""
import csv, glob, collections, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            key = k[k.index(pat):k.index(pat) + 40].split("(")[0]
            a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        print(k, {c: round(t / n) for c, (n, t) in sorted(v.items())})
