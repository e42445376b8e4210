#!/bin/bash
# Default-configuration kernel times of every bench workload.
cd "$(dirname "$0")/.."
for wl in "$@"; do echo "== $wl"; python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"],1), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'; done
