#!/bin/bash
# Kernel-time sweep of the fused kernel on one workload: tools/sweep_small.sh WORKLOAD "ENV=VAL ..." ...
WL=${1:-rr100k}; shift
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"],1), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'; }
run A=0
for spec in "$@"; do run $spec; done
