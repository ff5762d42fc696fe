#!/usr/bin/env python
"""bench.py — headline benchmark of the SPH hot path (SPHSystem::step) on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "simulation steps/sec ... DFSPH dam-break at stated N", config 5):
the 10,288,500-particle dam break (190 x 285 x 190 block, 920k boundary particles), DFSPHSolver
with fixed 1 divergence + 4 density iterations, dt = 0.002, fp32.  It fits one GPU, so N = 1 runs
the whole domain on one device and N > 1 splits the same domain into x-slabs (strong scaling).
A "step" is one SPHSystem::step(): neighbour search + solver step.  Inputs are resident in HBM
before the timed region (the scene is uploaded by the constructor).

Prints ONE JSON line (rank 0).  Extra legs: `roofline` (dominant kernel, live hipEvent timing over
the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
# algorithmic bytes per particle: whole DFSPH(v,d) step incl. neighbour search,
# and the density-error sweep alone (R pos12 vel12 mass4 density4 alpha4, W error4 kappa4)
def step_bytes_per_particle(v, d):
    return 420 + 92 * v + 104 * d + 72
RATE_KERNEL_BYTES_PER_PARTICLE = 44
DOMINANT_SPAN = "density_error"  # k_rate<DENSITY_MODE>: computeDensityError_CUDA, DFSPHSolver.cu:94-116


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=190, help="fluid block is nx x 1.5nx x nx (190 -> 10,288,500)")
    ap.add_argument("--solver", default="dfsph", choices=["wcsph", "dfsph", "pbd"])
    ap.add_argument("--div-iters", type=int, default=1)
    ap.add_argument("--den-iters", type=int, default=4)
    ap.add_argument("--pbd-iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="run the x-slab driver even with one rank (debug)")
    ap.add_argument("--cpu-nx", type=int, default=40, help="bounded CPU sample: nx of the oracle run")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(args, n_bench):
    """CPU oracle (kind 'port': this repo's restatement of the reference, OpenMP over particles) on a
    bounded sample of the same workload, scaled linearly in particle count to the bench size."""
    from oracle import oracle as O
    P, fluid, boundary = O.scene(args.cpu_nx)
    P.solver = {"wcsph": O.WCSPH, "dfsph": O.DFSPH, "pbd": O.PBD}[args.solver]
    P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = args.div_iters, args.den_iters, args.pbd_iters
    s = O.System(P, fluid, boundary)          # constructor step = warm-up
    t0 = time.time()
    for _ in range(args.cpu_steps):
        s.step()
    dt = (time.time() - t0) / args.cpu_steps
    n_s = len(fluid)
    cores = O.lib().oracle_max_threads()
    steps_per_s_at_bench = (1.0 / dt) * (n_s / float(n_bench))
    return {"value": steps_per_s_at_bench, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/sph_oracle.c, %s dam-break nx=%d (%d particles), %d steps at %.3f s/step on %d OpenMP "
                      "threads, scaled by particle count to %d particles" % (args.solver, args.cpu_nx, n_s, args.cpu_steps,
                                                                             dt, cores, n_bench)}


def read_traffic(workload_key):
    """per-launch HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload_key)
            return entry["hbm_bytes_per_launch"] if entry else None      # provenance: "source" / "measured_on" in that file
    except Exception:
        return None


def emit(result, real_stdout):
    """the ONE JSON line, on the process's original stdout"""
    os.write(real_stdout, (json.dumps(result) + "\n").encode())


def main():
    args = parse()
    # Everything but the result line goes to stderr: RCCL prints its version banner and the engine its
    # notices (e.g. PBD's first-step message, PBDSolver.cu:45-49) on the C-level stdout, buffered until
    # exit, i.e. after a Python print.  stdout carries exactly one line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")

    # torch ships its own HIP runtime (same SONAME as /opt/rocm's): import it first so that libsphx.so
    # binds to the SAME runtime instance and both sides see the device, streams and pointers
    import torch
    import sphx
    if sphx.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")

    if args.gpus > 1 or args.force_slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        from multi_gpu import run_slab_bench        # x-slab decomposition, torch.distributed (RCCL)
        result = run_slab_bench(args, rank, world, local_rank)
        if rank == 0:
            emit(result, real_stdout)
        return

    sphx.set_device(local_rank)
    torch.cuda.set_device(local_rank)

    P, fluid, boundary = sphx.scene(args.nx)
    solver = {"wcsph": sphx.WCSPH, "dfsph": sphx.DFSPH, "pbd": sphx.PBD}[args.solver]
    P.solver = solver
    P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = args.div_iters, args.den_iters, args.pbd_iters
    if solver == sphx.WCSPH:
        P.dt = 0.001
    sim = sphx.System(P, fluid, boundary)     # uploads + constructor step (SPHSystem.cu:69-76)
    n = sim.n
    if solver == sphx.PBD:
        sim.step()                            # PBD: the constructor step only records positions

    # warm-up (untimed; also captures the hipGraph used when the live timer is off)
    if args.warmup > 0:
        sim.step_n(args.warmup)
    torch.cuda.synchronize()

    # timed region: exactly K steps, launched back to back with one sync at the end; the dominant
    # kernel's launches are bracketed by hipEvents on the engine stream (live roofline leg)
    span = DOMINANT_SPAN if solver == sphx.DFSPH else ""
    if span:
        sphx.kernel_timer(True, span)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_events = sim.step_n(args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    spans = sphx.kernel_timer_collect() if span else {}
    sphx.kernel_timer(False)

    ms_per_step = wall * 1e3 / args.steps
    steps_per_s = args.steps / wall
    if solver == sphx.DFSPH:
        bpp = step_bytes_per_particle(args.div_iters, args.den_iters)
    elif solver == sphx.WCSPH:
        bpp = 396
    else:
        bpp = 300 + 104 * args.pbd_iters + 72
    result = {
        "metric": "simulation steps/sec, DFSPH dam-break" if solver == sphx.DFSPH else "simulation steps/sec, %s dam-break" % args.solver,
        "value": steps_per_s, "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "dam-break %dx%dx%d = %d fluid + %d boundary particles, %s%s, dt=%g"
                               % (args.nx, 3 * args.nx // 2, args.nx, n, sim.nb, args.solver.upper(),
                                  "(%d div + %d density iters, fixed)" % (args.div_iters, args.den_iters) if solver == sphx.DFSPH
                                  else ("(%d Jacobi iters)" % args.pbd_iters if solver == sphx.PBD else ""), P.dt),
                   "particles": n, "decomposition": "single device",
                   "step_algorithmic_bytes_per_particle": bpp,
                   "step_algorithmic_GBps": bpp * n * steps_per_s / 1e9,
                   "step_hbm_roofline_frac": bpp * n * steps_per_s / 1e9 / HBM_PEAK_GBPS,
                   "event_ms_per_step": ms_events / args.steps},
    }
    if span and span in spans:
        tot_ms, launches = spans[span]
        avg_ms = tot_ms / launches
        achieved = RATE_KERNEL_BYTES_PER_PARTICLE * n / (avg_ms * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": "k_rate<DENSITY_MODE> (span '%s')" % span,
                              "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBPS,
                              "traffic": read_traffic("%s_nx%d" % (args.solver, args.nx)),
                              "avg_launch_ms": avg_ms, "launches": launches,
                              "algorithmic_bytes_per_launch": RATE_KERNEL_BYTES_PER_PARTICLE * n}
    else:
        result["roofline"] = None
    sim.close()
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, n)
    emit(result, real_stdout)


if __name__ == "__main__":
    main()
