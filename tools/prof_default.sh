#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench line (rr1m, or $2).  Usage: tools/prof_default.sh outdir [workload]
out=${1:-gpurun_out/prof}
wl=${2:-rr1m}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$out/trace" -- \
    python3 "$ROOT/bench.py" --workload $wl --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-parity-mode > "$ROOT/$out/bench_under_rocprof_$wl.json" 2> /dev/null
find "$ROOT/$out/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/$out/rocprofv3_kernel_stats_bench_$wl.csv"
rm -rf "$ROOT/$out/trace"
cut -d, -f1-4 "$ROOT/$out/rocprofv3_kernel_stats_bench_$wl.csv" | sed 's/void (anonymous namespace):://; s/(.*)"/"/; s/_ZN12_GLOBAL__N_1[0-9]*//' | cut -c1-90 | head -8
