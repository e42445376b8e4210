#!/usr/bin/env python3
"""Benchmark of the layout iteration (BASELINE.json metric: layout iterations/s at fixed
n_vertices, with achieved fraction of roofline), one process per GPU.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one layout iteration (spring + KNN + intersection + integrate/normalise) of the
whole graph.  Default workload: random-regular n=1,000,000 d=8 (E=4,000,000), n_components=3,
n_neighbors=10, sample_size=256 -- the graph BASELINE.json quotes its target on.  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Order of a run: engine + positions; the device is brought to its steady power state (spin_up: the same iterations, untimed,
for 150 ms -- `spinup_steps` in the JSON line -- after which the starting positions are put back); W untimed warm-up steps;
`--repeats` timed passes of EXACTLY K steps each, barrier + synchronize on both sides, the median reported (`value`,
`ms_per_step`; every pass in `ms_per_step_passes`); then a pass with HIP-event timers for the per-kernel table, the
parity-mode record and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, n, param, D, k, S)
    "rr1m": ("rr", 1_000_000, 8, 3, 10, 256),      # the >=10x target case (BASELINE.json north_star)
    "rr100k": ("rr", 100_000, 8, 3, 10, 256),      # BASELINE configs[1]
    "er1m": ("er", 1_000_000, 1e-5, 3, 10, 256),   # BASELINE configs[2]
    "rr4m": ("rr", 4_000_000, 8, 3, 10, 256),      # BASELINE configs[3] (meant for 8 GPUs)
    "snap16": ("er", 4039, 0.0108, 16, 32, 256),   # BASELINE configs[4] shape (facebook_combined size), D=16 k=32
    "rr20k": ("rr", 20_000, 8, 3, 10, 256),        # quick self-test
    "rr1m_d6": ("rr", 1_000_000, 8, 6, 10, 256),   # wide rows at scale (8-float rows)
    "rr1m_d12": ("rr", 1_000_000, 8, 12, 10, 256), # (16-float rows, 32-deep contraction)
    # SURVEY 8d's streaming point: a 256 MB position table (the size of the Infinity Cache), B_iter = 2.75 GB: the
    # gathers of the spring phase come from HBM, not from cache
    "rr16m": ("rr", 16_000_000, 8, 3, 10, 256),
    # a graph WITH structure (SNAP-like): 1000 communities of 1000 vertices, 8 neighbours inside + 2 anywhere, vertex numbers
    # shuffled -- what an internal vertex order can and cannot do for the gathers (DESIGN.md "Vertex order")
    "pp1m": ("pp", 1_000_000, (1000, 8, 2), 3, 10, 256),
    "pp1m_sorted": ("pp", 1_000_000, (1000, 8, 2, 0, False), 3, 10, 256),   # the same graph in community order: what the best vertex order could give
}
HBM_PEAK = 8.0e12       # B/s  (MI355X_MICROARCH.md: HBM3E peak, spec)
FP32_PEAK = 157.3e12    # FLOP/s (fp32 vector peak = dense fp32 MFMA peak)


def make_workload(name, seed=0):
    import graphem_rapids_amd as gra
    kind, n, prm, D, k, S = WORKLOADS[name]
    # (the big graphs take minutes to generate: kept in /tmp for the later passes of a profiling job on the same box)
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"graphem_bench_{name}_seed{seed}.npy")
    if n >= 4_000_000 and os.path.exists(cache):
        edges = np.load(cache)
    else:
        if kind == "rr":
            edges = gra.random_regular_edges(n, prm, seed=seed)
        elif kind == "pp":
            edges = gra.planted_partition_edges(n, *prm[:3], seed=seed, shuffle=(prm[4] if len(prm) > 4 else True))
        else:
            edges = gra.erdos_renyi_edges(n, prm, seed=12345)
        edges = np.ascontiguousarray(edges, dtype=np.int32)
        if n >= 4_000_000:
            try:
                np.save(cache, edges)
            except OSError:
                pass
    rng = np.random.default_rng(seed)
    pos = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)  # the reference's own random start (pt.py:369)
    return n, D, k, S, edges, pos


def pmc_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 correction applied; tools/profile_round.sh writes the file).
    bench.py cannot run the profiler on itself, so this is the latest committed measurement, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "final", f"traffic_{workload}.json")))
    if not files:
        return None
    rec = json.load(open(files[-1]))
    if rec.get("kernel") != kernel:
        return None
    rec["file"] = os.path.relpath(files[-1], ROOT)
    return rec


def cpu_baseline(n, D, k, S, edges, pos, budget_s=24.0):
    """The reference's algorithm on this host's cores, three ways, each on a bounded number of iterations of the SAME
    workload: `port_omp` = the C oracle with every phase over all cores (oracle.OmpStepper), `port` = the plain oracle
    (KNN phase OpenMP, other phases one thread, the order-exact restatement the parity tests use), `torch_cpu` = a
    PyTorch-CPU restatement of the reference's own torch ops without its MemoryManager / gc.collect wrappers
    (oracle/torch_cpu.py).  `value` is the FASTEST of them."""
    import oracle
    import torch
    from oracle import torch_cpu
    rng = np.random.default_rng(1)
    samples = [rng.permutation(len(edges))[:S].astype(np.int32) for _ in range(5)]

    def timed(step, share):
        # a leg whose iteration is cheap against its share runs one untimed iteration first (page faults, cold caches, the
        # OpenMP team's start: 5 iterations of 25 ms are too few to hide them); a slow leg (seconds per iteration) counts it
        p = pos.copy()
        t0 = time.perf_counter()
        p = step(p, samples[0])
        t_first = time.perf_counter() - t0
        warm = t_first < share / 8
        done, t_total = (0, 0.0) if warm else (1, t_first)
        i = 1
        while done < len(samples) and (done == 0 or t_total + t_total / done < share):
            t0 = time.perf_counter()
            p = step(p, samples[i % len(samples)])
            t_total += time.perf_counter() - t0
            done += 1
            i += 1
        return {"value": done / t_total, "ms_per_iter": 1e3 * t_total / done, "iterations": done, "warmup_iterations": 1 if warm else 0}

    omp = oracle.OmpStepper(n, edges)
    legs = {}
    all_threads = oracle.num_threads()
    best_omp = None
    for nt in sorted({all_threads, max(1, all_threads // 2), min(all_threads, 96), min(all_threads, 64), min(all_threads, 32), min(all_threads, 16)}, reverse=True):
        oracle.set_threads(nt)   # memory-bound phases on a multi-socket host: the best team is not always the largest
        leg = dict(timed(lambda p, s: omp.step(p, s, k), budget_s / 12), cores=nt,
                   what="oracle/graphem_oracle.c go_step_omp: every phase OpenMP; best of several team sizes")
        if best_omp is None or leg["value"] > best_omp["value"]:
            best_omp = leg
    oracle.set_threads(all_threads)
    legs["port_omp"] = best_omp
    legs["port"] = dict(timed(lambda p, s: oracle.step(p, edges, s, k), budget_s / 3), cores=all_threads,
                        what="oracle/graphem_oracle.c go_step: KNN phase OpenMP, other phases 1 thread")
    te = torch.from_numpy(edges.astype(np.int64))

    def tstep(p, s):
        with torch.no_grad():
            return torch_cpu.step(torch.from_numpy(p), te, torch.from_numpy(s.astype(np.int64)), k).numpy()
    legs["torch_cpu"] = dict(timed(tstep, budget_s / 3), cores=torch.get_num_threads(),
                             what="oracle/torch_cpu.py: cdist + topk, index_add_, unbiased std; no MemoryManager")
    best = max(legs, key=lambda name: legs[name]["value"])
    return {"value": legs[best]["value"], "unit": "iterations/s", "cores": legs[best]["cores"], "kind": "port",
            "which": best, "nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "sample": f"{legs[best]['iterations']} full iterations of the same workload per leg; value = the fastest leg ({best})",
            "ms_per_iter": legs[best]["ms_per_iter"], "port_omp": legs["port_omp"], "port": legs["port"],
            "torch_cpu": legs["torch_cpu"]}


def fused_roofline(kern, E_rank, n_rank, D, workload, world):
    """(roofline, roofline_knn_fp32_equivalent) of the dominant kernel from its HIP-event duration.  SURVEY 8d, the parts
    of B_iter the fused spring+scan kernel performs: spring (edge list, read pos, zero F, write F), KNN midpoints (edge
    list, read pos) and the un-normalised update (read pos, read F, write new) = 16 E + 7 * (n D 4) bytes per launch."""
    dom = "spring_scan" if "spring_scan" in kern else "spring_mid" if "spring_mid" in kern else "knn_scan"
    scan_us = kern.get(dom, {}).get("avg_us")
    if not scan_us:
        return None, None
    b_alg = 16.0 * E_rank + 28.0 * n_rank * D if dom == "spring_scan" else 8.0 * E_rank + 4.0 * n_rank * D
    ach = b_alg / (scan_us * 1e-6)
    roofline = {"kernel": dom, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": ach / HBM_PEAK, "traffic": None, "avg_launch_us": scan_us,
                "algorithmic_bytes_per_launch": b_alg,
                "random_row_fetches_per_launch": 2.0 * E_rank,
                "row_fetch_rate_G_per_s": 2.0 * E_rank / (scan_us * 1e-6) / 1e9,
                "row_fetch_ceiling_G_per_s": 73.9,  # tools/micro/gather_bench.hip: 8M random 16-B rows of a 16 MB table in 108 us
                }
    if world == 1 and workload:
        rec = pmc_traffic(workload, dom)
        if rec:
            tb = rec["traffic_bytes"]
            roofline["traffic"] = tb
            roofline["traffic_source"] = rec["file"]
            roofline["traffic_measured_at_commit"] = rec.get("commit")
            roofline["traffic_note"] = ("PMC counters (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc passes, read side doubled per "
                                        "the guide's gfx950 note) of this kernel in the committed profile named in traffic_source -- "
                                        "bench.py cannot profile itself, so NOT measured in this run")
            if tb:
                roofline["traffic_rate_GBps"] = tb / (scan_us * 1e-6) / 1e9
                roofline["traffic_frac_of_peak"] = tb / (scan_us * 1e-6) / HBM_PEAK
    # the same kernel's arithmetic side (F_knn = 3 D S E per launch) as an fp32-EQUIVALENT rate: the
    # algorithm's fp32 flops over the kernel's duration against the fp32 vector peak.  NOT a pipe utilisation: the
    # pre-filter runs on the f16 matrix pipe (split-f16 operands) and spends far fewer instructions per pair.
    return roofline, scan_us


def iter_roofline(ms, E, n, D):
    b_iter = 16.0 * E + 36.0 * n * D                       # B_iter = 16E + 36nD
    return {"bound": "hbm", "achieved": b_iter / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": b_iter / (ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_iter": b_iter, "traffic": None}


def cold_pass(run, sync, args):
    """ms per step of the FIRST W + K steps this process runs on the engine, before spin_up: what a caller sees on a device
    that idled while the host built the graph (clock ramp; one-off launches of the first iteration are in the W steps)."""
    run(args.warmup)
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    return 1e3 * (time.perf_counter() - t0) / args.steps


def cpu_baseline_pinned(args):
    """cpu_baseline() in a child process whose OpenMP threads are PINNED (OMP_PROC_BIND=close, OMP_PLACES=cores, unless the
    caller's environment says otherwise).  Unpinned, the OpenMP port's memory-bound phases lose half their rate to thread
    migration on the GPU box's 2-socket host and the figure moves 21-35 it/s from box to box (rr1m, same box, back to back:
    25.3 / 23.8 unpinned, 25.3 spread over cores, 36.2 spread over hardware threads, 48.1 close over cores --
    tools/cpu_baseline_probe.py).  A child, because libgomp reads the placement when it is loaded (torch loads it) and because
    pinning this process would pin the host sampler's thread onto the enqueuing thread's core."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload]
    if args.dim:
        cmd += ["--dim", str(args.dim)]
    if args.sample_size:
        cmd += ["--sample-size", str(args.sample_size)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    line = next((ln for ln in reversed(out.stdout.splitlines()) if ln.startswith("{")), None)
    if out.returncode != 0 or line is None:
        raise RuntimeError(f"cpu baseline child failed ({out.returncode}): {out.stderr[-500:]}")
    rec = json.loads(line)
    rec["omp_env"] = {"OMP_PROC_BIND": env["OMP_PROC_BIND"], "OMP_PLACES": env["OMP_PLACES"]}
    return rec


def parity_mode(args, n, D, k, S, edges, pos, device_id):
    """The same K steps on an engine in the mode whose neighbour rows ARE the reference's (knn_distance='cdist': torch.cdist's
    values, torch.topk's tie order, pt.py:580-583; ids drawn on the host like torch.randperm's, before the timed region):
    median of the same number of passes, per-kernel HIP-event durations from a further pass."""
    from graphem_rapids_amd import _native
    E = len(edges)
    eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, device_id=device_id, knn_distance="cdist")
    eng.set_positions(pos)
    rng = np.random.default_rng(3)
    stream = np.stack([rng.permutation(E)[:S] for _ in range(max(args.steps, args.warmup))]).astype(np.int32) if S < E else None
    run = lambda iters: eng.run(iters, None if stream is None else stream[:iters])
    cold_ms = cold_pass(run, eng.sync, args)
    eng.set_positions(pos)
    spin_up(run, eng.sync, max(1, args.steps), lambda: eng.set_positions(pos))
    run(args.warmup)
    passes = []
    for _ in range(max(1, args.repeats)):
        eng.sync()
        t0 = time.perf_counter()
        run(args.steps)
        eng.sync()
        passes.append(time.perf_counter() - t0)
    dt = sorted(passes)[len(passes) // 2]
    listed = []
    eng.timing_enable(True)
    eng.timing_reset()
    run(args.steps)
    eng.sync()
    timings = eng.timings()
    eng.timing_enable(False)
    for _ in range(8):   # how many rows get partial_sort's heap replayed (ties among the k + 2 smallest values)
        eng.step(None if stream is None else stream[0])
        listed.append(int(eng.knn_cdist_stats()[0]))
    eng.close()
    kern = {name: {"avg_us": 1e3 * tot / cnt, "launches_per_step": cnt / args.steps}
            for name, (tot, cnt) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
    ms = 1e3 * dt / args.steps
    return {"knn_distance": "cdist", "sampler": "host", "reference_identical": True,
            "what": "neighbour rows = the reference's torch.cdist + torch.topk rows id for id (pt.py:580-583); ids handed in like torch.randperm's",
            "value": args.steps / dt, "unit": "iterations/s",
            "ms_per_step": ms, "ms_per_step_passes": [1e3 * p / args.steps for p in passes],
            "ms_per_step_cold": cold_ms,
            "replayed_rows_per_step_sample": listed,
            "roofline": fused_roofline(kern, E, n, D, None, None)[0],
            "roofline_iter_hbm": iter_roofline(ms, E, n, D),
            "kernels": kern}


def public_api(args, device_id):
    """The seam the north star names, timed the way the reference times itself (graphem_rapids/benchmark.py:131-133: wall
    clock around run_layout): create_graphem(adjacency, D, backend='hip', init='random', ...).run_layout(K), INCLUDING the
    final download of the positions, after one untimed run_layout(W) -- on BASELINE configs[1] (rr100k) and on the bench
    workload, with sampler='torch' (the parity default up to 2**20 edges: ids = torch.randperm(E)[:S] of the global CPU
    generator, knn_distance='cdist') and sampler='device' (the speed mode).  `torch_randperm_ms` = what ONE
    torch.randperm(E) call costs on this host: the per-iteration host work of this path until round 4."""
    import torch
    import graphem_rapids_amd as gra
    from graphem_rapids_amd import _native
    K = max(args.steps, 100)   # run_layout's own default length (pt.py:808) unless more steps were asked for
    out = {"protocol": f"wall clock around run_layout({K}) incl. the download of the positions; run_layout({args.warmup}) before, untimed; "
                       "median of 3", "steps": K, "host_twist": _native.torch_randperm_isa(), "cases": {}}
    for wl in dict.fromkeys(["rr100k", args.workload]):
        n, D, k, S, edges, pos = make_workload(wl)
        adj = gra.edges_to_adjacency(n, edges)
        E = len(edges)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.randperm(E)
        rp_ms = 1e3 * (time.perf_counter() - t0) / 3
        st = torch.get_rng_state().numpy().copy()
        draw_us = float("inf")
        for _ in range(3):   # (the fastest of three: a descheduled millisecond on a shared host reads as 3 ms per draw)
            t0 = time.perf_counter()
            _native.torch_randperm_prefix(st, E, S, 20)
            draw_us = min(draw_us, 1e6 * (time.perf_counter() - t0) / 20)
        rec = {"n_vertices": n, "n_edges": E, "torch_randperm_ms": rp_ms, "host_draw_us_per_iteration": draw_us}
        for sampler in ("torch", "device"):
            t0 = time.perf_counter()
            emb = gra.create_graphem(adj, n_components=D, backend="hip", init="random", n_neighbors=k, sample_size=S,
                                     sampler=sampler, verbose=False, seed=0, device=f"cuda:{device_id}")
            t_create = time.perf_counter() - t0
            emb.positions = pos
            emb.run_layout(max(1, args.warmup))
            spin_up(emb.run_layout, emb._engine.sync, 20, lambda: setattr(emb, "positions", pos))
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                res = emb.run_layout(K)
                walls.append(time.perf_counter() - t0)
            assert res.shape == (n, D) and np.isfinite(res).all()
            t0 = time.perf_counter()
            emb.get_positions()
            t_dl = time.perf_counter() - t0
            w = sorted(walls)[1]
            host = None
            if sampler == "torch":
                st = emb._engine.sampler_stats()
                host = {k: float(v) for k, v in st.items()}
            rec[sampler] = {"host_sampler_ms_last_run": host, "knn_distance": emb.knn_distance, "reference_identical": emb.knn_distance == "cdist" and sampler == "torch",
                            "wall_ms": 1e3 * w, "ms_per_iteration": 1e3 * w / K, "iterations_per_s": K / w,
                            "wall_ms_passes": [1e3 * x for x in walls], "download_ms": 1e3 * t_dl, "create_s": t_create}
            del emb
        out["cases"][wl] = rec
    return out


SPINUP_MS = 150.0


def spin_up(run, sync, steps, reset, agree=None):
    """The device's power state, not the workload: an MI355X that sat idle while the host built the graph runs its first
    ~15 ms of kernels at lower clocks (rr1m: 20-step passes of 177, 175, 169, 167, 166 us per iteration after 3 s of idle
    against 164 back to back on a busy device, the SAME layout either way -- tools/warm_probe.py, profiles/r04/final/warm_probe.log).  So the same iterations
    run untimed for SPINUP_MS first, then `reset` puts the starting positions back: the W warm-up steps and the K timed
    steps that follow start from the layout they would have started from without this, on a device in its steady state
    (the engine's iteration counter, which keys the device sampler, is not put back: same layout, other sample draws).
    The pass BEFORE this is reported as `ms_per_step_cold`."""
    # (every rank must run the SAME number of iterations -- they contain collectives: one timed round, the slowest rank's
    # time agreed on, the number of further rounds derived from that)
    run(steps)   # (the first round carries one-off launches: not the one to measure)
    sync()
    t0 = time.perf_counter()
    run(steps)
    sync()
    dt = agree(time.perf_counter() - t0) if agree else time.perf_counter() - t0
    rounds = int(min(2000, max(0, np.ceil(SPINUP_MS * 1e-3 / max(dt, 1e-6)) - 1)))
    for _ in range(rounds):
        run(steps)
    sync()
    reset()
    return (2 + rounds) * steps


def self_launch(args):
    """One rank per GPU through torch.distributed.run, supervised: the child's JSON line is passed through; a native-loop
    attempt that fails or hangs is ended (its whole process group) and repeated with the Python-driven loop."""
    import signal
    import socket
    import subprocess

    def attempt(loop, limit_s):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        argv = [a for a in sys.argv[1:]]
        if "--loop" in argv:
            i = argv.index("--loop")
            del argv[i:i + 2]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv + ["--loop", loop]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)   # the session this call started: torchrun and its ranks, nothing else
            proc.wait()
            return None, f"no result after {limit_s} s"
        line = next((ln for ln in reversed(out.splitlines()) if ln.startswith("{")), None)
        if proc.returncode != 0 or line is None:
            return None, f"exit code {proc.returncode}"
        return line, None

    line, why = attempt(args.loop, 900)
    if line is None and args.loop == "native":
        sys.stderr.write(f"bench.py: native loop failed ({why}); repeating with the Python-driven loop\n")
        line, why = attempt("python", 900)
    if line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed: {why}\n")
        return 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rr1m", choices=sorted(WORKLOADS))
    ap.add_argument("--sampler", default="device", choices=["device", "host"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of cpu_baseline_pinned()
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the parity_mode sub-record (the same steps with knn_distance='cdist')")
    ap.add_argument("--no-public-api", action="store_true", help="skip the public_api sub-record (wall clock around create_graphem(...).run_layout(K))")
    ap.add_argument("--sample-size", type=int, default=None, help="override the workload's number of sampled midpoints")
    ap.add_argument("--knn", default="auto", choices=["auto", "scan", "grid", "ivf"],
                    help="KNN search (gh_params.knn_method); ivf is approximate (recall: tools/ivf_probe.py)")
    ap.add_argument("--ivf-lists", type=int, default=0)
    ap.add_argument("--ivf-probes", type=int, default=0)
    ap.add_argument("--knn-distance", default="exact", choices=["exact", "cdist"],
                    help="exact = speed mode; cdist = the reference's cdist + topk rows (parity mode, gh_params.knn_distance)")
    ap.add_argument("--dim", type=int, default=None, help="override the workload's number of components (experiments)")
    ap.add_argument("--dist", action="store_true",
                    help="use the multi-GPU driver (RCCL collectives) even for one rank: rehearsal of the N>1 path")
    ap.add_argument("--loop", default="python", choices=["python", "native"],
                    help="N > 1: 'python' drives each iteration's kernels and collectives through torch.distributed (RCCL); "
                         "'native' = gh_run_partitioned, the loop inside the C library over its own RCCL communicator "
                         "(opt-in until a world > 1 run has passed on hardware)")
    ap.add_argument("--finish", default="overlap", choices=["overlap", "own", "gathered"],
                    help="N > 1: how an iteration ends (graphem_rapids_amd/distributed.py).  overlap = form D: the all-gather of the "
                         "un-normalised rows starts right after the fused kernel, on a second stream / communicator, and runs beside "
                         "select -> keys -> merge -> statistics; every rank normalises all n rows.  own = form C (round 4): three "
                         "collectives in a row, the rows last.  gathered = form B")
    ap.add_argument("--repeats", type=int, default=3, help="timed passes of --steps iterations; the median is reported")
    args = ap.parse_args()

    if args.cpu_baseline_only:   # child of cpu_baseline_pinned(): no GPU, no torch.distributed; one JSON line
        n, D, k, S, edges, pos = make_workload(args.workload)
        if args.dim:
            D = args.dim
            pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
        if args.sample_size:
            S = min(args.sample_size, len(edges))
        print(json.dumps(cpu_baseline(n, D, k, S, edges, pos)))
        return

    # `python bench.py --gpus N` without a launcher: start the N ranks here, as children, BEFORE anything of this process
    # touches a GPU (a process that has initialised HIP must not be replaced or re-executed; torch is not even imported yet)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Rehearsal of the N > 1 code path on a box with ONE GPU (tests, development): GRAPHEM_BENCH_REHEARSAL=gloo puts every
    # rank on cuda:0 and runs the collectives over gloo (RCCL refuses two ranks on one device).  Exercises everything of a
    # multi-rank run but RCCL; its timings mean nothing.
    rehearsal = os.environ.get("GRAPHEM_BENCH_REHEARSAL") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    from graphem_rapids_amd import _native
    n, D, k, S, edges, pos = make_workload(args.workload)
    if args.dim:
        D = args.dim
        pos = (np.random.default_rng(0).standard_normal((n, D)) * 0.1).astype(np.float32)
    E = len(edges)
    if args.sample_size:
        S = min(args.sample_size, E)

    use_dist = world > 1 or args.dist
    if use_dist:
        # a multi-rank run that stalls (a collective some rank never enters) would sit until the launcher's limit: after 20
        # minutes every rank prints where it is and leaves, so that the log of a failed scaling run says why
        import faulthandler
        faulthandler.dump_traceback_later(1200, exit=True)
        import torch.distributed as dist
        from graphem_rapids_amd.distributed import PartitionedLayout
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        lay = PartitionedLayout(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, rank=rank, world=world,
                                device_id=local_rank, native=args.loop == "native", knn_distance=args.knn_distance, finish=args.finish)
        lay.set_positions(pos)
        run = lay.run
        sync = lay.sync
        barrier = dist.barrier
        eng = lay.engine.eng
    else:
        eng = _native.Engine(n, D, edges, 1.0, 0.2, 0.5, k, S, seed=0, device_id=local_rank, knn_method=args.knn,
                             knn_distance=args.knn_distance, ivf_lists=args.ivf_lists, ivf_probes=args.ivf_probes)
        eng.set_positions(pos)
        stream = None
        if args.sampler == "host":  # ids drawn on the host before the timed region (parity-style stream)
            rng = np.random.default_rng(3)
            stream = np.stack([rng.permutation(E)[:S] for _ in range(max(args.steps, args.warmup))]).astype(np.int32)

        def run(iters):
            eng.run(iters, None if stream is None else stream[:iters])
        sync = eng.sync

        def barrier():
            return None

    def agree(dt_local):   # the slowest rank's time, the same number on every rank
        if not use_dist:
            return dt_local
        t = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    reset = (lambda: lay.set_positions(pos)) if use_dist else (lambda: eng.set_positions(pos))
    cold_ms = agree(cold_pass(run, sync, args))
    reset()
    spinup_steps = spin_up(run, sync, max(1, args.steps), reset, agree)
    run(args.warmup)
    passes = []
    for _ in range(max(1, args.repeats)):   # median of 3 repeats; every pass is exactly --steps iterations
        sync()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        sync()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        passes.append(dt)
    dt = sorted(passes)[len(passes) // 2]

    # per-kernel durations with HIP events on the launching stream: a second pass of the same K steps
    eng.timing_enable(True)
    eng.timing_reset()
    if use_dist:
        lay.time_overlap = True
    run(args.steps)
    sync()
    timings = eng.timings()
    eng.timing_enable(False)
    exposed_py = None
    if use_dist:
        lay.time_overlap = False
        if lay.exposed_events:   # Python-driven form D: how long the engine's stream still waited for the early all-gather
            torch.cuda.synchronize()
            exposed_py = 1e3 * sum(a.elapsed_time(b) for a, b in lay.exposed_events) / len(lay.exposed_events)
            lay.exposed_events.clear()

    if rank == 0:
        ms = 1e3 * dt / args.steps
        kern = {name: {"avg_us": 1e3 * tot / cnt, "launches_per_step": cnt / args.steps}
                for name, (tot, cnt) in sorted(timings.items(), key=lambda kv: -kv[1][0])}
        E_rank = E // world
        n_rank = n // world
        # Dominant kernel: the fused spring+scan kernel (the stand-alone scan when unfused).  It is bound by the
        # memory system: the spring phase gathers one position row per pull-list entry (2E random 16-byte rows),
        # and with no locality in the graph most of them miss the L2 (PMC traffic below).
        roofline, scan_us = fused_roofline(kern, E_rank, n_rank, D, args.workload, world)
        knn_fp32 = None
        if scan_us:
            flops_scan = 3.0 * D * S * E_rank
            knn_fp32 = {"kernel": roofline["kernel"], "bound": "mfma", "kind": "fp32-equivalent rate, not pipe utilisation",
                        "what": "the search's algorithmic fp32 flops (3 D S E per launch) over the kernel's duration against the fp32 vector "
                                "peak; the work itself is done by a split-f16 pre-filter on v_mfma_f32_32x32x16_f16 (D <= 3; packed fp32 VALU "
                                "otherwise) plus an exact fp32 re-check of the few pairs that pass, in far fewer instructions per pair",
                        "achieved": flops_scan / (scan_us * 1e-6) / 1e12, "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s (fp32-equivalent)",
                        "frac": flops_scan / (scan_us * 1e-6) / FP32_PEAK, "algorithmic_flops_per_launch": flops_scan}
        hbm = iter_roofline(ms, E, n, D)
        out = {
            "metric": "layout iterations/s", "value": args.steps / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps": spinup_steps, "ms_per_step": ms,
            "ms_per_step_cold": cold_ms,
            "repeats": len(passes), "ms_per_step_passes": [1e3 * p / args.steps for p in passes],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "graph": WORKLOADS[args.workload][0], "n_vertices": n,
                       "n_edges": E, "n_components": D, "n_neighbors": k, "sample_size": S,
                       "sampler": args.sampler, "knn": args.knn, "knn_distance": args.knn_distance,
                       "parallelism": f"rows/{world}" + (("+gloo-rehearsal-on-one-gpu" if rehearsal else "+rccl") if use_dist else ""),
                       # speed mode (knn_distance='exact', device sampler): exact-difference distances, ties on the smaller id --
                       # 252-256 of 256 neighbour rows equal the reference's at 1 M vertices; the mode whose rows ARE the
                       # reference's on every vertex is the `parity_mode` record of this line (reference_identical: true)
                       "reference_identical": args.knn_distance == "cdist" and args.sampler == "host"},
            "roofline": roofline, "roofline_knn_fp32": knn_fp32, "roofline_iter_hbm": hbm, "kernels": kern,
        }
        if use_dist:  # rank 0's split of an iteration (HIP events on the engine's stream, second pass): where a scaling run loses its time
            per = lambda name: kern[name]["avg_us"] * kern[name]["launches_per_step"] if name in kern else 0.0
            comp = sum(v["avg_us"] * v["launches_per_step"] for name, v in kern.items() if not name.startswith("allgather"))
            # collectives on the engine's stream are exposed in full; form D's early all-gather of the rows runs on a side
            # stream ("allgather_rows": its whole duration there) and only what the engine's stream then still waits for is
            # exposed ("allgather_rows_exposed", native loop; the Python-driven loop measures the same wait with CUDA events)
            on_stream = sum(per(nm) for nm in kern if nm.startswith("allgather") and nm not in ("allgather_rows", "allgather_rows_exposed"))
            rows_total = per("allgather_rows")
            native = bool(getattr(lay, "native", False))
            if args.finish == "overlap":
                # (native loop without a second communicator -- no ncclCommSplit --: the rows went out on the engine's stream)
                rows_exposed = (per("allgather_rows_exposed") if "allgather_rows_exposed" in kern else rows_total) if native else exposed_py
            else:
                rows_exposed = rows_total
            out["rank0_us_per_step"] = {
                "kernels": comp, "collectives_exposed": on_stream + (rows_exposed or 0.0),
                "small_collectives_on_stream": on_stream, "rows_allgather_total": rows_total if native or args.finish != "overlap" else None,
                "rows_allgather_exposed": rows_exposed,
                "rows_allgather_hidden": (rows_total - rows_exposed) if (rows_exposed is not None and rows_total) else None,
                "finish": args.finish,
                "note": "HIP events, second pass of the same K steps (events add launch gaps).  No N > 1 run of this code on hardware "
                        "existed when it was written (development boxes have one GPU): this record is what a multi-GPU driver run fills in",
                "loop": "gh_run_partitioned (C library; RCCL all-gathers, form D's rows on a second stream + ncclCommSplit communicator)"
                        if native else "python-driven (torch.distributed; form D's rows on a side stream + a process group of their own)"}
            out["config"]["finish"] = args.finish
        if world == 1 and not use_dist and args.knn_distance == "exact" and not args.no_parity_mode:
            out["parity_mode"] = parity_mode(args, n, D, k, S, edges, pos, local_rank)
        if world == 1 and not use_dist and not args.no_public_api:
            out["public_api"] = public_api(args, local_rank)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_pinned(args)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if use_dist:
        import faulthandler
        import torch.distributed as dist
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
