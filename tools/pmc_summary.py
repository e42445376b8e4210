"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter values per dispatch."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()):
        x = x[len(x) // 2:]  # skip warm-up dispatches
        print(f"    {c:28s} {sum(x) / len(x):16.0f}")
