#!/bin/bash
# End-of-round measurements of round 5 on the GPU box.  Usage: GRAPHEM_COMMIT=<hash> tools/final_round_r05.sh outdir
# (the driver's bench command, the other workloads, parity-mode lines, world-1 rehearsals of the partitioned loop over RCCL --
# form D native and Python-driven, form C --, the float64 engine, kernel traces of the small workloads)
out=${1:-gpurun_out/final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command_rr1m.json 2> $out/bench_driver_command_rr1m.err || exit 1
echo "driver command done"
for wl in rr100k er1m rr4m snap16 rr1m_d6 rr1m_d12 pp1m; do
  python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-public-api > $out/bench_$wl.json 2>/dev/null || echo "$wl failed"
  echo "$wl done"
done
python bench.py --dist --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > $out/bench_dist_world1_python_overlap_rr1m.json 2>/dev/null || echo "dist python failed"
python bench.py --dist --loop native --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > $out/bench_dist_world1_native_overlap_rr1m.json 2>/dev/null || echo "dist native failed"
python bench.py --dist --loop native --finish own --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > $out/bench_dist_world1_native_own_rr1m.json 2>/dev/null || echo "dist native own failed"
python bench.py --dist --loop native --workload rr4m --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > $out/bench_dist_world1_native_overlap_rr4m.json 2>/dev/null || echo "dist native rr4m failed"
python bench.py --dist --loop native --knn-distance cdist --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-public-api > $out/bench_dist_world1_native_cdist_rr1m.json 2>/dev/null || echo "dist native cdist failed"
echo "dist done"
python tools/bench_f64.py rr100k rr1m > $out/bench_f64.log 2>&1
python tools/cdist_probe.py rr1m 30 > $out/cdist_probe_rr1m.log 2>&1
echo "f64, cdist probe done"
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "%.1f it/s" % d["value"], "%.1f us" % (1e3 * d["ms_per_step"]), "cold %.1f" % (1e3 * d.get("ms_per_step_cold", 0)), "frac %.4f" % (d["roofline"]["frac"] if d.get("roofline") else -1),
          {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
    if "parity_mode" in d:
        p = d["parity_mode"]; print("   parity: %.1f it/s %.1f us" % (p["value"], 1e3 * p["ms_per_step"]), p["replayed_rows_per_step_sample"], {k: round(v["avg_us"], 1) for k, v in p["kernels"].items()})
    if "public_api" in d:
        for w, c in d["public_api"]["cases"].items():
            print("   public_api", w, "torch %.1f us/it" % (1e3 * c["torch"]["ms_per_iteration"]), "device %.1f us/it" % (1e3 * c["device"]["ms_per_iteration"]), "draw %.1f us" % c["host_draw_us_per_iteration"], "torch.randperm %.2f ms" % c["torch_randperm_ms"])
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("   cpu:", c["which"], "%.2f it/s" % c["value"], {k: round(c[k]["value"], 2) for k in ("port_omp", "port", "torch_cpu")}, "cores", c["nproc"])
    if "rank0_us_per_step" in d:
        print("   rank0:", {k: v for k, v in d["rank0_us_per_step"].items() if k not in ("note", "loop")})
PY
