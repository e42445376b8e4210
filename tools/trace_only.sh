#!/bin/bash
# rocprofv3 kernel-trace summary of one bench command (no counters): tools/trace_only.sh <out.csv> <bench args ...>
OUTCSV=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_only
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_only -- \
    python3 "$ROOT/bench.py" "$@" --repeats 1 --no-cpu-baseline --no-parity-mode --no-public-api > /tmp/trace_only.json 2>/dev/null
find /tmp/trace_only -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/$OUTCSV"
grep "^{" /tmp/trace_only.json | tail -1 > "$ROOT/${OUTCSV%.csv}_bench_line.json"
