#!/bin/bash
# Sweep of the threshold-subset stride on the bench workload (GRAPHEM_HIP_SUBSET_STRIDE).  Usage: tools/stride_sweep.sh out_prefix workload strides...
out=$1; wl=$2; shift 2
for s in "$@"; do
  GRAPHEM_HIP_SUBSET_STRIDE=$s python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > ${out}_${wl}_s${s}.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads(open("${out}_${wl}_s${s}.json").read().strip().splitlines()[-1])
print("stride $s", "%.1f us" % (1e3*d["ms_per_step"]), {k: round(v["avg_us"],1) for k,v in d["kernels"].items()})
PY
done
