#!/bin/bash
# End-of-round measurements on the GPU box: bench lines of every BASELINE workload, the world-1 rehearsal of the
# partitioned loop, the grid/scan sweep.  Usage: tools/final_round.sh outdir
out=${1:-gpurun_out/final}
mkdir -p $out
python bench.py --steps 50 --warmup 5 > $out/bench_default_rr1m.json 2> $out/bench_default_rr1m.err || exit 1
for wl in rr100k er1m rr4m snap16; do
  python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$wl.json 2>/dev/null || echo "$wl failed"
done
python bench.py --dist --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_dist_world1_rr1m.json 2>/dev/null || echo "dist failed"
python bench.py --dist --workload rr4m --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_dist_world1_rr4m.json 2>/dev/null || echo "dist rr4m failed"
tools/knn_method_sweep.sh $out/km rr1m 256 1024 4096 8192 16384 > $out/knn_method_sweep.log 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "%.1f it/s" % d["value"], "%.1f us" % (1e3 * d["ms_per_step"]), "frac %.3f" % (d["roofline"]["frac"] if d.get("roofline") else -1),
          {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("   cpu:", c["which"], "%.2f it/s" % c["value"], {k: round(c[k]["value"], 2) for k in ("port_omp", "port", "torch_cpu")}, "cores", c["nproc"])
PY
cat $out/knn_method_sweep.log
