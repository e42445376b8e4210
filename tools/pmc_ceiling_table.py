"""Table of the counter passes of tools/pmc_ceiling.sh: one column per configuration (gather micro-benchmark at four table
sizes, the fused kernel on rr1m), one row per counter: mean per dispatch of the kernel of interest (second half of its
dispatches: the first ones are warm-ups)."""
import collections, csv, glob, os, re, sys
raw = sys.argv[1]
cols = collections.OrderedDict()
for d in sorted(glob.glob(raw + "/*")):
    name = os.path.basename(d)
    cfg = re.sub(r"_g\d+$", "", name)
    want = "gather<0>" if cfg.startswith("gather") else "spring_scan"
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"].replace(" ", "") or (want == "gather<0>" and re.search(r"gather<\s*0\s*>", r["Kernel_Name"])):
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, x in vals.items():
        x = x[len(x) // 2:]
        cols.setdefault(cfg, {})[c] = sum(x) / len(x)
names = sorted({c for v in cols.values() for c in v})
order = sorted(cols, key=lambda c: (not c.startswith("gather"), int(re.sub(r"\D", "", c) or 0)))
print(f"{'counter':40s}" + "".join(f"{c:>18s}" for c in order))
for c in names:
    print(f"{c:40s}" + "".join(f"{cols[o].get(c, float('nan')):18.4g}" for o in order))
