#!/bin/bash
# Fused-kernel time across vertex orders and tile mappings: tools/order_sweep.sh <workload> ...
b() { python bench.py --no-cpu-baseline --repeats 1 --steps 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' %8.1f it/s %7.1f us  fused %6.1f us' % (d['value'], 1e3*d['ms_per_step'], d['kernels']['spring_scan']['avg_us']))"; }
for wl in "$@"; do
  for ro in 1 2 3; do for xm in 0 1; do
    echo -n "$wl reorder=$ro xcd_map=$xm "; GRAPHEM_HIP_REORDER=$ro GRAPHEM_HIP_XCD_MAP=$xm b --workload $wl
  done; done
done
