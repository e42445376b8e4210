#!/bin/bash
# SQ + TCC counters of the fused kernel for one workload under the current environment: tools/pmc_quick.sh <outdir> <workload>
OUT=$1; WL=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES \
    --output-format csv -d "$ROOT/$OUT/pmc_sq" -- python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum \
    --output-format csv -d "$ROOT/$OUT/pmc_tcc" -- python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum \
    --output-format csv -d "$ROOT/$OUT/pmc_tcp" -- python3 "$ROOT/tools/run_knn_only.py" "$WL" 8 run > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/pmc_sq" "$ROOT/$OUT/pmc_tcc" "$ROOT/$OUT/pmc_tcp" | grep -A18 spring_scan > "$ROOT/$OUT/pmc_fused.txt"
rm -rf "$ROOT/$OUT"/pmc_sq "$ROOT/$OUT"/pmc_tcc "$ROOT/$OUT"/pmc_tcp
cat "$ROOT/$OUT/pmc_fused.txt"
