#!/bin/bash
# Parity-mode (knn_distance = cdist) timings on the GPU box: plain bench lines (rr1m, rr100k) and the rocprofv3 kernel-trace
# summary of the rr1m run.  Usage: tools/cdist_round.sh outdir
out=${1:-gpurun_out/cdist}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/$out
cd $ROOT
python bench.py --knn-distance cdist --sampler host --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_cdist_rr1m.json 2>/dev/null || echo "cdist rr1m failed"
python bench.py --knn-distance cdist --sampler host --workload rr100k --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_cdist_rr100k.json 2>/dev/null || echo "cdist rr100k failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/trace" -- \
    python3 "$ROOT/bench.py" --knn-distance cdist --sampler host --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-parity-mode > "$ROOT/$out/bench_cdist_under_rocprof.json" 2> /dev/null
find "$ROOT/$out/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/$out/rocprofv3_kernel_stats_cdist_rr1m.csv"
rm -rf "$ROOT/$out/trace"
cd $ROOT
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "%.1f it/s" % d["value"], "%.1f us" % (1e3 * d["ms_per_step"]), {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
cut -d, -f1-7 $out/rocprofv3_kernel_stats_cdist_rr1m.csv | sed 's/void (anonymous namespace):://; s/(.*)"/"/' | head -12
