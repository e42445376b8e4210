#!/bin/bash
# Which unit saturates when 16-byte rows are gathered at random?  Counter passes (TCP / TA / TCC / EA) of the bare gather
# micro-benchmark (tools/micro/gather_bench.hip, 8 M gathers, plain loads) at table sizes 8 / 16 / 64 / 256 MB and of the
# fused spring+scan kernel on rr1m (16 MB table, 8 M gathers + the scan).  One rocprofv3 --pmc run per counter group and
# configuration (counters of one block share a few slots); summary: tools/pmc_ceiling_table.py.
# Usage: tools/pmc_ceiling.sh <outdir>
set -u
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/$OUT"
BIN=/tmp/gather_bench
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$ROOT/tools/micro/gather_bench.hip" -o $BIN || exit 1
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"
 "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
 "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_CYCLE_sum"
 "GRBM_GUI_ACTIVE TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_REQ_sum"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"
 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"
)
# plain timings first (no profiler)
: > "$ROOT/$OUT/gather_times.log"
for ROWS in 500000 1000000 4000000 16000000; do
    timeout -k 10 120 $BIN $ROWS 1000000 0 >> "$ROOT/$OUT/gather_times.log" 2>&1 || exit 1
done
cat "$ROOT/$OUT/gather_times.log"
g=0
for GRP in "${GROUPS_[@]}"; do
    for ROWS in 500000 1000000 4000000 16000000; do
        timeout -k 10 120 rocprofv3 --pmc $GRP --output-format csv -d "$ROOT/$OUT/raw/gather_${ROWS}_g$g" -- $BIN $ROWS 1000000 0 > /dev/null 2>"$ROOT/$OUT/err_gather_${ROWS}_g$g.log" \
            || echo "group $g rows $ROWS failed" >> "$ROOT/$OUT/failed.log"
    done
    timeout -k 10 300 rocprofv3 --pmc $GRP --output-format csv -d "$ROOT/$OUT/raw/fused_g$g" -- python3 "$ROOT/tools/run_knn_only.py" rr1m 8 run > /dev/null 2>"$ROOT/$OUT/err_fused_g$g.log" \
        || echo "group $g fused failed" >> "$ROOT/$OUT/failed.log"
    g=$((g + 1))
    echo "group $g done"
done
python3 "$ROOT/tools/pmc_ceiling_table.py" "$ROOT/$OUT/raw" > "$ROOT/$OUT/pmc_ceiling_table.txt"
find "$ROOT/$OUT" -name "err_*.log" -size 0 -delete
rm -rf "$ROOT/$OUT/raw"
cat "$ROOT/$OUT/pmc_ceiling_table.txt"
