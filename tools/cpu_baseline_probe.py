"""bench.py's cpu_baseline on a workload under the current OpenMP environment (thread placement experiments).
python tools/cpu_baseline_probe.py [workload]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "rr1m"
n, D, k, S, edges, pos = bench.make_workload(wl)
r = bench.cpu_baseline(n, D, k, S, edges, pos)
print(os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), "->", round(r["value"], 2), "it/s, best", r["which"], "cores", r["cores"],
      {k2: round(r[k2]["value"], 2) for k2 in ("port_omp", "port", "torch_cpu")}, flush=True)
