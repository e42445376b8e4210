#!/bin/bash
# Iteration time of the filtered scan against the grid (n_components <= 3) and IVF searches over the number of sampled
# midpoints.  IVF is approximate: its recall at the same settings comes from tools/ivf_probe.py.
# Usage: tools/knn_method_sweep.sh <workload> <dim or 0> S1 S2 ...
wl=${1:-rr1m}; dim=${2:-0}; shift 2
extra=""; [ "$dim" != "0" ] && extra="--dim $dim"
for S in "$@"; do
  methods="scan grid ivf ivfx"; [ "$dim" != "0" ] && [ "$dim" -gt 3 ] && methods="scan ivf ivfx"
  for m in $methods; do
    knn="--knn $m"; [ "$m" = "ivfx" ] && knn="--knn ivf --ivf-probes -1"   # ivfx: the inverted file in its exact mode
    python bench.py --workload $wl $extra --sample-size $S $knn --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl dim=$dim S=$S $m: %9.1f us/iter ' % (1e3*d['ms_per_step']), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})"
  done
done
