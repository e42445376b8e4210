#!/bin/bash
# Kernel-trace summary and one SQ counter pass of the inverted-file path (AUTO at thousands of queries = its exact mode).
# Usage: tools/profile_ivf.sh <outdir> <workload> <dim or 0> <sample size>
set -u
OUT=$1; WL=$2; DIM=$3; S=$4
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
EXTRA=""; [ "$DIM" != "0" ] && EXTRA="--dim $DIM"
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- \
    python3 "$ROOT/bench.py" --workload "$WL" $EXTRA --sample-size "$S" --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-parity-mode \
    > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES \
    --output-format csv -d "$ROOT/$OUT/pmc_sq" -- \
    python3 "$ROOT/bench.py" --workload "$WL" $EXTRA --sample-size "$S" --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
find "$ROOT/$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/$OUT/kernel_stats.csv"
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
with open(out + "/pmc_sq_summary.txt", "w") as o:
    o.write("per-launch averages (SQ counters; *_CYCLES of waves in quad-cycles, MFMA busy in cycles)\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        n = max(cnt[k], 1)
        o.write(f"{k:60s} launches {n:4d}  " + "  ".join(f"{c}={v[c] / n:.3g}" for c in sorted(v)) + "\n")
PY
rm -rf "$ROOT/$OUT"/trace "$ROOT/$OUT"/pmc_sq
echo "profile written to $OUT"
