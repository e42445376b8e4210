#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace summary of the benchmark command and
# separate PMC passes for HBM traffic of the dominant kernel.  Usage: tools/profile_round.sh <outdir> [workload]
set -u
OUT=${1:-gpurun_out/profile}
WL=${2:-rr1m}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- \
    python3 "$ROOT/bench.py" --workload "$WL" --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/bench_under_rocprof.err"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_$c" -- \
        python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES \
    --output-format csv -d "$ROOT/$OUT/pmc_sq" -- python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum \
    --output-format csv -d "$ROOT/$OUT/pmc_tcc" -- python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
find "$ROOT/$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/$OUT/kernel_stats.csv"
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/pmc_FETCH_SIZE" "$ROOT/$OUT/pmc_WRITE_SIZE" "$ROOT/$OUT/pmc_sq" "$ROOT/$OUT/pmc_tcc" > "$ROOT/$OUT/pmc_summary.txt"
python3 "$ROOT/tools/make_traffic_json.py" "$ROOT/$OUT" "$WL" > /dev/null
echo "profile written to $OUT"
