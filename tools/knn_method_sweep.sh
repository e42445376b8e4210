#!/bin/bash
# scan vs grid KNN over the number of sampled midpoints.  Usage: tools/knn_method_sweep.sh out_prefix workload S...
out=$1; wl=$2; shift 2
for S in "$@"; do
  for m in scan grid; do
    python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --sample-size $S --knn $m > ${out}_${wl}_S${S}_${m}.json 2>/dev/null || { echo "S=$S $m failed"; continue; }
    python - <<PY
import json
d=json.loads(open("${out}_${wl}_S${S}_${m}.json").read().strip().splitlines()[-1])
print("S=$S $m", "%.1f us" % (1e3*d["ms_per_step"]), {k: round(v["avg_us"],1) for k,v in d["kernels"].items()})
PY
  done
done
