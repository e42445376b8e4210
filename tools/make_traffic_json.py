"""Per-launch HBM-side traffic of the dominant kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
TCC counters), with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md spelled out.
Usage: make_traffic_json.py <profile dir made by tools/profile_round.sh> <workload> [kernel substring]"""
import collections, csv, glob, json, os, sys
root, workload = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "spring_scan"
vals = collections.defaultdict(list)
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq", "pmc_tcc"):
    for f in glob.glob(f"{root}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: (lambda x: sum(x[len(x) // 2:]) / len(x[len(x) // 2:]))(v) for k, v in vals.items()}  # skip warm-up launches
fetch_kb, write_kb = mean.get("FETCH_SIZE", 0.0), mean.get("WRITE_SIZE", 0.0)
out = {
    "workload": workload, "kernel": kernel,
    "commit": os.environ.get("GRAPHEM_COMMIT"),   # the commit the profiled tree was built from (handed in by the caller: the GPU box has no .git)
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean per launch; tools/profile_round.sh",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "traffic_bytes_uncorrected": (fetch_kb + write_kb) * 1024.0,
    "traffic_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so the "
                  "read side is doubled; the calibration is for wide coalesced reads, this kernel's reads are 16-byte row "
                  "gathers (one 128-byte line each), so the corrected figure is an upper bound and the uncorrected one a "
                  "lower bound; Infinity-Cache hits are included (the position table is cache resident)",
}
for k in ("TCC_EA0_RDREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
    if k in mean:
        out[k] = mean[k]
json.dump(out, open(f"{root}/traffic_{workload}.json", "w"), indent=1)
print(json.dumps(out))
