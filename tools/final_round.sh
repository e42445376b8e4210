#!/bin/bash
# End-of-round measurements on the GPU box.  Usage: tools/final_round.sh outdir   (bench lines of every workload, parity-mode
# lines, the world-1 rehearsals of the partitioned loop, spectral-init timings, the full-size parity logs)
out=${1:-gpurun_out/final}
mkdir -p $out
python bench.py --steps 50 --warmup 5 > $out/bench_default_rr1m.json 2> $out/bench_default_rr1m.err || exit 1
for wl in rr100k er1m rr4m rr16m snap16 rr1m_d6 rr1m_d12 pp1m; do
  python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_$wl.json 2>/dev/null || echo "$wl failed"
done
python bench.py --knn-distance cdist --sampler host --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_cdist_rr1m.json 2>/dev/null || echo "cdist failed"
python bench.py --knn-distance cdist --sampler host --workload rr100k --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_cdist_rr100k.json 2>/dev/null || echo "cdist 100k failed"
python bench.py --dist --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_dist_world1_python_rr1m.json 2>/dev/null || echo "dist python failed"
python bench.py --dist --loop native --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_dist_world1_native_rr1m.json 2>/dev/null || echo "dist native failed"
python bench.py --dist --loop native --workload rr4m --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_dist_world1_native_rr4m.json 2>/dev/null || echo "dist native rr4m failed"
(python tools/bench_init.py 100000; python tools/bench_init.py 1000000) > $out/bench_init.log 2>&1
python tools/cdist_probe.py rr1m 30 > $out/cdist_probe_rr1m.log 2>&1
python tools/bench_f64.py rr100k rr1m > $out/bench_f64.log 2>&1
python bench.py --dist --loop native --knn-distance cdist --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode > $out/bench_dist_world1_native_cdist_rr1m.json 2>/dev/null || echo "dist native cdist failed"
python -m pytest tests/test_hip_reference_fullsize.py tests/test_hip_f64.py tests/test_spectral_init.py -q -m gpu -s > $out/gpu_tests_fullsize_f64_spectral.log 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "%.1f it/s" % d["value"], "%.1f us" % (1e3 * d["ms_per_step"]), "frac %.4f" % (d["roofline"]["frac"] if d.get("roofline") else -1),
          {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("   cpu:", c["which"], "%.2f it/s" % c["value"], {k: round(c[k]["value"], 2) for k in ("port_omp", "port", "torch_cpu")}, "cores", c["nproc"])
PY
