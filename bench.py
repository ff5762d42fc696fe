#!/usr/bin/env python
"""bench.py — headline benchmark of the SPH hot path (SPHSystem::step) on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "simulation steps/sec ... DFSPH dam-break at stated N", config 5):
the 10,288,500-particle dam break (190 x 285 x 190 block, 917k boundary particles), DFSPHSolver
with fixed 1 divergence + 4 density iterations, dt = 0.002, fp32.  It fits one GPU, so N = 1 runs
the whole domain on one device and N > 1 splits the same domain into x-slabs (strong scaling).
A "step" is one SPHSystem::step(): neighbour search + solver step.  Inputs are resident in HBM
before the timed region (the scene is uploaded by the constructor).

Prints ONE JSON line (rank 0) with, besides the contract's keys:
  value / roofline   the headline leg: --arith persistent by default (the north star's contract: results within 1e-5 of the reference,
                     integer cell indices bit-exact; the reference's own binary is built -use_fast_math; neighbour rows kept across steps
                     while a device-side check allows) -- dominant kernel (density-error
                     sweep) timed live with hipEvents over the timed region: HBM fraction from algorithmic bytes, FP32-VALU fraction from
                     counted pair evaluations, PMC traffic from profiles/traffic.json when that file was measured on THIS source tree
  legs_by_arithmetic the same workload and window under the other contracts (strict = bit-exact IEEE, tolerance = the same arithmetic as
                     the headline with rows rebuilt every step), each with its own live roofline block
  reference_default_iterations   the same scene and window under the reference's own adaptive iteration control (strict and headline arithmetic)
  steady_state       post-impact legs (ragged cells, wall contact): the 10 M scene under the reference's adaptive iteration
                     control and the 1 M config 3, >= 100 timed steps each, with neighbours-per-particle statistics
  configs            BASELINE configs 2, 3, 4 (263k WCSPH, 1M DFSPH, 1M PBD) and the reference scene (20,736 particles)
  cpu_baseline       the CPU oracle: BASELINE config 1 and the reference scene's other two solvers with 1 thread and with all cores
                     (measured), the headline solver at 1 M on all cores (extrapolated to the bench size, labelled); rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector (packed) peak
# algorithmic bytes per particle: whole step incl. neighbour search, and the density-error sweep
# alone (R pos12 vel12 mass4 density4 alpha4, W error4 kappa4)
RATE_KERNEL_BYTES_PER_PARTICLE = 44
# flop model of SURVEY.md §8d for one accepted pair of the rate sweep: gradW ~ 40 (1 sqrt + divisions counted as 1
# each) + (v_i - v_j).gradW, mass factor and accumulation ~ 10
RATE_KERNEL_FLOP_PER_PAIR = 50
DOMINANT_SPAN = "density_error"  # k_rate_quad<DENSITY_MODE> (k_rate<> without rows): computeDensityError_CUDA, DFSPHSolver.cu:94-116


def step_bytes_per_particle(solver, v, d, k, fixed=True):
    if solver == "dfsph":
        # (with FIXED counts the 44-byte error sweep behind the last divergence correction has no reader and is
        # not launched -- DFSPHSolver::step -- so it is not counted either)
        return 420 + 92 * v + 104 * d + 72 - (44 if (fixed and v >= 1) else 0)
    if solver == "wcsph":
        return 396
    return 300 + 104 * k + 72


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=190, help="fluid block is nx x 1.5nx x nx (190 -> 10,288,500)")
    ap.add_argument("--solver", default="dfsph", choices=["wcsph", "dfsph", "pbd"])
    ap.add_argument("--div-iters", type=int, default=1)
    ap.add_argument("--den-iters", type=int, default=4)
    ap.add_argument("--pbd-iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the steady-state leg and the configs 2-4 legs")
    ap.add_argument("--settle-steps", type=int, default=300, help="steps advanced before the steady-state leg is timed")
    ap.add_argument("--steady-steps", type=int, default=100)
    ap.add_argument("--force-slab", action="store_true", help="run the x-slab layer with one process: --slabs loopback slabs on one device")
    ap.add_argument("--slabs", type=int, default=1, help="with --force-slab: number of loopback slabs")
    ap.add_argument("--slab-transport", default="loopback", choices=["loopback", "rccl"],
                    help="with --force-slab: device-to-device copies, or the installed RCCL (grouped sends to self on the communication stream)")
    ap.add_argument("--no-overlap", action="store_true", help="slab layer: stage-then-exchange instead of edge-first stages")
    ap.add_argument("--arith", default="persistent", choices=["strict", "tolerance", "persistent"],
                    help="arithmetic contract of the HEADLINE leg: persistent (default: the north star's 1e-5 contract -- the reference's own "
                         "binary is built -use_fast_math -- with neighbour rows kept across steps), tolerance (the same arithmetic, rows rebuilt "
                         "every step) or strict (bit-exact IEEE, the parity contract)")
    ap.add_argument("--tuning", default="", help="engine tuning for experiments: comma-separated sphx_tuning fields, e.g. row_capacity=64,dfsph_no_tail=1 "
                                                  "(include/sphx_c.h; the library reads no SPHX_* environment variables)")
    ap.add_argument("--cpu-nx", type=int, default=88, help="bounded CPU sample: nx of the oracle run (88 -> 1,022,208)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-full-steps", type=int, default=2,
                    help="steps of the CPU oracle at the BENCH size itself (10.3 M particles: ~17 s per step on 128 threads); 0: extrapolate from --cpu-nx")
    return ap.parse_args()


def cpu_baseline(args, n_bench):
    """CPU oracle (kind 'port': this repo's restatement of the reference, OpenMP over particles; the reference itself
    cannot be built here -- nvcc, Thrust and helper_math.h are absent) on bounded samples:
      * BASELINE config 1 as stated (20,736-particle WCSPH dam break, dt = 0.001) and the same scene under the reference's
        default DFSPH and PBD(20), each with ONE thread (the north star's "serial kernels") and with all cores -- measured,
        nothing extrapolated (timer position: SPHSystem.cu:131-157, one step() per frame; README.md:6-9 quotes GPU frame times);
      * the headline solver at the bench size itself on all cores (r06: --cpu-full-steps steps behind the constructor step; measured,
        not extrapolated), with the 1 M sample of the earlier rounds beside it (`configs`)."""
    from oracle import oracle as O
    cores = O.lib().oracle_max_threads()
    configs = []
    for name, solver, dt, serial_steps, parallel_steps in (("wcsph", O.WCSPH, 0.001, 20, 200), ("dfsph", O.DFSPH, 0.002, 6, 60), ("pbd", O.PBD, 0.002, 6, 60)):
        for threads, steps in ((1, serial_steps), (cores, parallel_steps)):
            P, fluid, boundary = O.scene(24)
            P.solver = solver; P.dt = dt
            s = O.System(P, fluid, boundary, threads=threads)      # constructor step = warm-up
            if solver == O.PBD:
                s.step()
            t0 = time.time()
            for _ in range(steps):
                s.step()
            sec = (time.time() - t0) / steps
            s.close()
            configs.append({"workload": "reference scene, 20,736 particles, %s, dt=%g%s" % (
                                {"wcsph": "WCSPH (BASELINE config 1)", "dfsph": "DFSPH adaptive (reference defaults)", "pbd": "PBD(20)"}[name], dt,
                                ", free fall" ), "threads": threads, "steps": steps, "ms_per_step": sec * 1e3, "steps_per_s": 1.0 / sec,
                            "extrapolated": False})
            note("cpu baseline: reference scene %s, %d thread(s): %.1f ms/step" % (name, threads, sec * 1e3))
    P, fluid, boundary = O.scene(args.cpu_nx)
    P.solver = {"wcsph": O.WCSPH, "dfsph": O.DFSPH, "pbd": O.PBD}[args.solver]
    P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = args.div_iters, args.den_iters, args.pbd_iters
    note("cpu baseline: oracle nx=%d" % args.cpu_nx)
    s = O.System(P, fluid, boundary, threads=cores)          # constructor step = warm-up
    t0 = time.time()
    for _ in range(args.cpu_steps):
        s.step()
    dt = (time.time() - t0) / args.cpu_steps
    s.close()
    n_s = len(fluid)
    configs.append({"workload": "%s dam-break nx=%d, %d particles (the headline solver settings)" % (args.solver, args.cpu_nx, n_s),
                    "threads": cores, "steps": args.cpu_steps, "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "extrapolated": False})
    if args.cpu_full_steps > 0 and args.nx != args.cpu_nx:
        P, fluid, boundary = O.scene(args.nx)
        P.solver = {"wcsph": O.WCSPH, "dfsph": O.DFSPH, "pbd": O.PBD}[args.solver]
        P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = args.div_iters, args.den_iters, args.pbd_iters
        note("cpu baseline: oracle at the bench size, nx=%d (%d particles): constructor step + %d steps" % (args.nx, len(fluid), args.cpu_full_steps))
        s = O.System(P, fluid, boundary, threads=cores)
        t0 = time.time()
        for _ in range(args.cpu_full_steps):
            s.step()
        dtf = (time.time() - t0) / args.cpu_full_steps
        s.close()
        note("cpu baseline: %.2f s/step at the bench size" % dtf)
        configs.append({"workload": "%s dam-break nx=%d, %d particles (the bench size, the headline solver settings)" % (args.solver, args.nx, len(fluid)),
                        "threads": cores, "steps": args.cpu_full_steps, "ms_per_step": dtf * 1e3, "steps_per_s": 1.0 / dtf, "extrapolated": False})
        return {"value": 1.0 / dtf, "unit": "steps/s", "cores": cores, "kind": "port",
                "measured_steps_per_s": 1.0 / dtf, "measured_particles": len(fluid),
                "note": "port = oracle/sph_oracle.c, this repo's CPU restatement (OpenMP over particles); the reference is not buildable here",
                "sample": "%s dam-break nx=%d (%d particles = the bench size), %d steps behind the constructor step at %.2f s/step on %d OpenMP threads; "
                          "nothing extrapolated (1 M sample beside it: %.3f s/step, x %.2f per particle)" % (
                              args.solver, args.nx, len(fluid), args.cpu_full_steps, dtf, cores, dt, (dtf / len(fluid)) / (dt / n_s)),
                "configs": configs}
    steps_per_s_at_bench = (1.0 / dt) * (n_s / float(n_bench))
    return {"value": steps_per_s_at_bench, "unit": "steps/s", "cores": cores, "kind": "port",
            "extrapolated": "value = measured %.4f steps/s at %d particles x (%d / %d): linear in the particle count" % (1.0 / dt, n_s, n_s, n_bench),
            "measured_steps_per_s": 1.0 / dt, "measured_particles": n_s,
            "note": "port = oracle/sph_oracle.c, this repo's CPU restatement (OpenMP over particles); the reference is not buildable here",
            "sample": "%s dam-break nx=%d (%d particles), %d steps at %.3f s/step on %d OpenMP threads, scaled by "
                      "particle count to %d particles" % (args.solver, args.cpu_nx, n_s, args.cpu_steps, dt, cores, n_bench),
            "configs": configs}


def read_traffic(workload_key):
    """per-launch HBM bytes (and VALU-busy, when recorded) of the dominant kernel from the committed rocprofv3 --pmc
    passes — only if they were measured on the source tree this library was built from"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        from srchash import engine_source_hash
        with open(path) as f:
            entry = json.load(f).get(workload_key)
        if not entry or entry.get("source_hash") != engine_source_hash():
            return None
        return entry
    except Exception:
        return None


def limiter_text(pmc):
    """names both limits of the dominant kernel WITH the numbers measured on this source tree; without them it says so"""
    if not pmc or pmc.get("valu_issue_frac") is None:
        return "not measured for this build (profiles/traffic.json belongs to another source hash); r02 measurement: vector L1 / TA path"
    parts = ["VALU issue %.0f %% of the calibrated slots (SQ_INSTS_VALU x 2.3 clk, profiles/r03_valu_calibration.txt)" % (100.0 * pmc["valu_issue_frac"])]
    if pmc.get("l1_line_accesses_per_clk_per_cu") is not None:
        parts.append("vector L1 %.2f line accesses per clock per CU (ceiling ~1)" % pmc["l1_line_accesses_per_clk_per_cu"])
    if pmc.get("ta_busy_frac") is not None:
        parts.append("TA busy %.0f %%" % (100.0 * pmc["ta_busy_frac"]))
    if pmc.get("hbm_bytes_per_launch") and pmc.get("avg_launch_us_rocprof"):
        parts.append("HBM traffic %.2f TB/s of 8" % (pmc["hbm_bytes_per_launch"] / pmc["avg_launch_us_rocprof"] / 1e6))
    return "; ".join(parts)


_T0 = time.time()


def note(msg):
    """progress on stderr (stdout carries only the result line)"""
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - _T0, msg))
    sys.stderr.flush()


def emit(result, real_stdout):
    """the ONE JSON line, on the process's original stdout"""
    os.write(real_stdout, (json.dumps(result) + "\n").encode())


def neighbour_stats(sim):
    import numpy as np
    total, longest, hist = sim.row_stats()
    n = max(int(hist.sum()), 1)
    cdf = np.cumsum(hist) / float(n)
    return {"pairs": int(total), "mean": total / float(n), "p50": int(np.searchsorted(cdf, 0.5)),
            "p99": int(np.searchsorted(cdf, 0.99)), "max": int(longest)}


ARITH = {"strict": 0, "tolerance": 1, "persistent": 2}
ARITH_TEXT = {"strict": "strict arithmetic: every bit equals the IEEE evaluation of the reference's expressions in the reference's order (the parity contract "
                        "of tests/test_gpu_parity.py)",
              "tolerance": "tolerance arithmetic (sphx_params.reserved[3] = 1): v_rsq/v_rcp + FMA contraction in the neighbour sweeps, quad-per-particle row "
                           "walks with per-lane partial sums and one DPP reduction per particle; positions and densities within 1e-5 of the oracle, integer "
                           "fields bit-exact (tests/test_gpu_tolerance.py) -- the north star's contract; the reference's own binary is built -use_fast_math",
              "persistent": "tolerance arithmetic with persistent neighbour rows (reserved[3] = 2): the solver steps a working copy kept in row-build order, "
                            "rows carry a skin and are rebuilt when a device-side check finds > 0.49 skin of relative displacement; API arrays exported "
                            "in the reference's order every step (tests/test_gpu_tolerance.py::test_persistent_rows_*)"}


def make_system(sphx, nx, solver, div, den, pbd_iters, tolerance=False, arith=None):
    P, fluid, boundary = sphx.scene(nx)
    P.solver = {"wcsph": sphx.WCSPH, "dfsph": sphx.DFSPH, "pbd": sphx.PBD}[solver]
    P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = div, den, pbd_iters
    P.reserved[3] = ARITH[arith] if arith else (1 if tolerance else 0)
    if solver == "wcsph":
        P.dt = 0.001
    sim = sphx.System(P, fluid, boundary)     # uploads + constructor step (SPHSystem.cu:69-76)
    if solver == "pbd":
        sim.step()                            # PBD: the constructor step only records positions
    return sim, P


def timed_steps(torch, sim, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_events = sim.step_n(steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ms_events


def settled_leg(sphx, torch, nx, solver, div, den, settle, steps, arith="strict"):
    """advance a fresh system past the impact, then time `steps` steps there"""
    sim, P = make_system(sphx, nx, solver, div, den, 4, arith=arith)
    what = "DFSPH(%d,%d fixed)" % (div, den) if div >= 0 else "DFSPH(adaptive, reference defaults: 1e-3 thresholds, <= 20 iterations)"
    leg = {"workload": "dam-break nx=%d, %d particles, %s, dt=%g, %s arithmetic" % (nx, sim.n, what, P.dt, arith), "arithmetic": arith,
           "after_steps": 1 + settle}
    done, first = 0, None
    while done < settle:              # in slices, so that a run-away state shows up instead of eating the time budget
        k = min(50, settle - done)
        t0 = time.perf_counter()
        sim.step_n(k)
        done += k
        ms = (time.perf_counter() - t0) * 1e3 / k
        first = first or ms
        note("settling nx=%d: %d steps, %.2f ms/step" % (nx, done, ms))
        if ms > 25.0 * first:
            leg["diverged_after_steps"] = done
            sim.close()
            return leg
    wall, _ = timed_steps(torch, sim, steps)
    sps = steps / wall
    fixed = div >= 0
    if div < 0:
        div, den = sim.iters()
        leg["iterations_last_step"] = [div, den]
    bpp = step_bytes_per_particle(solver, div, den, 4, fixed)
    leg.update({"steps": steps, "steps_per_s": sps, "ms_per_step": wall * 1e3 / steps,
                "step_hbm_roofline_frac": bpp * sim.n * sps / 1e9 / HBM_PEAK_GBPS,
                "neighbours_per_particle": neighbour_stats(sim)})
    # what the ragged rows of this state cost the quad walk, and what bounding the walk at 48 entries (+ a compact launch for the tails)
    # could save at most
    rw = sim.row_walk_stats(48)
    rw["walked_over_even"] = rw["steps_walked"] / float(max(rw["steps_even_rows"], 1))
    rw["saving_of_a_cut_at_48"] = 1.0 - (rw["steps_cut"] + rw["steps_tail_launch"]) / float(max(rw["steps_walked"], 1))
    leg["row_walk"] = rw
    if arith == "persistent":
        # (in use = at the end of the leg: while nearly every step rebuilds its rows the controller leaves the mode for 256 steps at a time)
        in_use, builds, counted = sim.persistent_stats()
        leg["persistent_rows"] = {"in_use_at_end": in_use, "row_builds": builds, "steps_in_mode": counted}
    note("post-impact leg nx=%d (%s) done: %.2f ms/step" % (nx, arith, wall * 1e3 / steps))
    sim.close()
    return leg


def small_leg(sphx, torch, nx, solver, div, den, pbd_iters, steps, warmup, arith="strict", landed=0):
    """one of the BASELINE configs that are parity-test cases rather than the headline: graph-replayed steps.
    landed > 0 (the reference's own scene): the same number of steps is timed once more behind `landed` further steps, i.e. with the
    column on the floor -- the regime a run spends most of its frames in (adaptive DFSPH then runs 20 divergence iterations per step)"""
    sim, P = make_system(sphx, nx, solver, div, den, pbd_iters, arith=arith)
    sim.step_n(warmup)
    wall, _ = timed_steps(torch, sim, steps)
    note("leg nx=%d %s (%s): %.3f ms/step" % (nx, solver, arith, wall * 1e3 / steps))
    sps = steps / wall
    what = {"wcsph": "WCSPH", "dfsph": "DFSPH(%d,%d fixed)" % (div, den) if div >= 0 else "DFSPH(adaptive, reference defaults)",
            "pbd": "PBD(%d Jacobi)" % pbd_iters}[solver]
    fixed = div >= 0
    if solver == "dfsph" and div < 0:
        div, den = sim.iters()                  # iteration counts of the last step
        what += ", last step ran (%d,%d) iterations" % (div, den)
    bpp = step_bytes_per_particle(solver, div, den, pbd_iters, fixed)
    leg = {"workload": "dam-break nx=%d, %d particles, %s, dt=%g, %s arithmetic" % (nx, sim.n, what, P.dt, arith), "arithmetic": arith,
           "particles": sim.n, "steps": steps, "steps_per_s": sps, "ms_per_step": wall * 1e3 / steps,
           "algorithmic_GBps": bpp * sim.n * sps / 1e9, "hbm_roofline_frac": bpp * sim.n * sps / 1e9 / HBM_PEAK_GBPS}
    if landed > 0:
        sim.step_n(landed)
        wall2, _ = timed_steps(torch, sim, steps)
        leg["landed"] = {"behind_steps": warmup + steps + landed, "steps": steps, "ms_per_step": wall2 * 1e3 / steps, "steps_per_s": steps / wall2}
        if solver == "dfsph" and not fixed:
            leg["landed"]["last_step_iterations"] = list(sim.iters())
        note("leg nx=%d %s (%s), landed: %.3f ms/step" % (nx, solver, arith, wall2 * 1e3 / steps))
    sim.close()
    return leg


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
    if "SPHX_BENCH_DEVICE" in os.environ:        # probes that put several ranks on one device
        local_rank = int(os.environ["SPHX_BENCH_DEVICE"])
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")

    # torch ships its own HIP runtime (same SONAME as /opt/rocm's): import it first so that libsphx.so
    # binds to the SAME runtime instance and both sides see the device, streams and pointers
    import torch
    import sphx
    if sphx.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    if args.tuning:
        fields = dict(kv.split("=", 1) for kv in args.tuning.split(","))
        sphx.set_tuning(**{k: (float(v) if k == "pbd_skin" else int(v)) for k, v in fields.items()})

    if args.gpus > 1 or args.force_slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        from multi_gpu import run_slab_bench        # x-slab decomposition over RCCL
        result = run_slab_bench(args, rank, world, local_rank)
        if rank == 0:
            emit(result, real_stdout)
        return

    sphx.set_device(local_rank)
    torch.cuda.set_device(local_rank)
    solver = args.solver
    span = DOMINANT_SPAN if solver == "dfsph" else ""
    bpp = step_bytes_per_particle(solver, args.div_iters, args.den_iters, args.pbd_iters)
    what = {"dfsph": "DFSPH(%d div + %d density iters, fixed)" % (args.div_iters, args.den_iters),
            "pbd": "PBD(%d Jacobi iters)" % args.pbd_iters, "wcsph": "WCSPH"}[solver]

    def headline_leg(arith):
        """one arithmetic contract on the headline workload: W untimed warm-up steps, then exactly K timed steps launched back
        to back with one sync at the end; the dominant kernel's launches are bracketed by hipEvents on the engine stream (live
        roofline leg).  With the event timer on, the steps are launched eagerly rather than replayed from the captured hipGraph:
        at 10 M particles launch overhead is < 1 %, and the number stays the contract's "measured over the timed region"."""
        note("building the %s nx=%d scene (%s)" % (solver, args.nx, arith))
        sim, P = make_system(sphx, args.nx, solver, args.div_iters, args.den_iters, args.pbd_iters, arith=arith)
        n = sim.n
        if args.warmup > 0:
            sim.step_n(args.warmup)
        torch.cuda.synchronize()
        if span:
            sphx.kernel_timer(True, span)
        wall, ms_events = timed_steps(torch, sim, args.steps)
        spans = sphx.kernel_timer_collect() if span else {}
        sphx.kernel_timer(False)
        variant, kernel_name = sphx.last_rate_kernel()          # the instantiation the error sweeps of this leg were really launched as
        note("%s leg: %.2f ms/step" % (arith, wall * 1e3 / args.steps))
        nb = neighbour_stats(sim)
        sps = args.steps / wall
        leg = {"arithmetic": ARITH_TEXT[arith], "steps_per_s": sps, "ms_per_step": wall * 1e3 / args.steps, "event_ms_per_step": ms_events / args.steps,
               "step_algorithmic_GBps": bpp * n * sps / 1e9, "step_hbm_roofline_frac": bpp * n * sps / 1e9 / HBM_PEAK_GBPS,
               "particles": n, "boundary_particles": sim.nb, "dt": P.dt, "neighbours_per_particle": nb, "roofline": None}
        if arith == "persistent":
            in_use, builds, counted = sim.persistent_stats()
            leg["persistent_rows"] = {"in_use": in_use, "row_builds": builds, "steps": counted}
        if span and span in spans:
            tot_ms, launches = spans[span]
            avg_ms = tot_ms / launches
            achieved = RATE_KERNEL_BYTES_PER_PARTICLE * n / (avg_ms * 1e-3) / 1e9
            # counters of the dominant kernel AS THIS LEG RUNS IT (profiles/traffic.json, one entry per arithmetic contract, collected by
            # tools/collect_round.sh with the same bench arguments); None unless measured on exactly this source tree
            pmc = read_traffic("%s_nx%d_%s" % (solver, args.nx, arith))
            flops = RATE_KERNEL_FLOP_PER_PAIR * nb["pairs"] / (avg_ms * 1e-3) / 1e12
            leg["roofline"] = {"bound": "hbm", "kernel": "%s (span '%s')" % (kernel_name, span),
                               "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                               "avg_launch_ms": avg_ms, "launches": launches,
                               "algorithmic_bytes_per_launch": RATE_KERNEL_BYTES_PER_PARTICLE * n,
                               "valu": {"pairs_per_launch": nb["pairs"], "flop_per_pair_model": RATE_KERNEL_FLOP_PER_PAIR,
                                        "achieved_TFLOPs": flops, "peak_TFLOPs": FP32_PEAK_TFLOPS, "frac": flops / FP32_PEAK_TFLOPS,
                                        "valu_issue_frac_pmc": pmc.get("valu_issue_frac") if pmc else None,
                                        "valu_instr_per_simd_per_clk_pmc": pmc.get("valu_instr_per_simd_per_clk_raw") if pmc else None},
                               "limiter": limiter_text(pmc)}
        sim.close()
        return leg

    head = headline_leg(args.arith)
    n = head["particles"]
    result = {
        "metric": "simulation steps/sec, DFSPH dam-break" if solver == "dfsph" else "simulation steps/sec, %s dam-break" % solver,
        "value": head["steps_per_s"], "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "dam-break %dx%dx%d = %d fluid + %d boundary particles, %s, dt=%g, %s arithmetic, free-fall / early-impact window"
                               % (args.nx, 3 * args.nx // 2, args.nx, n, head["boundary_particles"], what, head["dt"], args.arith),
                   "arithmetic": head["arithmetic"],
                   "particles": n, "decomposition": "single device",
                   "step_algorithmic_bytes_per_particle": bpp,
                   "step_algorithmic_GBps": head["step_algorithmic_GBps"],
                   "step_hbm_roofline_frac": head["step_hbm_roofline_frac"],
                   "event_ms_per_step": head["event_ms_per_step"],
                   "neighbours_per_particle": head["neighbours_per_particle"]},
        "roofline": head["roofline"],
    }
    if "persistent_rows" in head:
        result["config"]["persistent_rows"] = head["persistent_rows"]

    if not args.no_extra_legs:
        # the same workload and window under the other arithmetic contracts, each with its own live roofline leg
        result["legs_by_arithmetic"] = {args.arith: {k: head[k] for k in ("steps_per_s", "ms_per_step", "step_hbm_roofline_frac")}}
        for other in ("strict", "tolerance", "persistent"):
            if other != args.arith:
                leg = headline_leg(other)
                result["legs_by_arithmetic"][other] = {k: leg[k] for k in ("arithmetic", "steps_per_s", "ms_per_step", "step_hbm_roofline_frac", "roofline") if k in leg}
                if "persistent_rows" in leg:
                    result["legs_by_arithmetic"][other]["persistent_rows"] = leg["persistent_rows"]
        # The reference's DEFAULT iteration control on the same scene and window (adaptive loops, thresholds 1e-3, at most 20 iterations --
        # main.cpp's DFSPHSolver; 1 divergence + 2 density iterations in free fall): loops decided on the device, the iterations beyond a
        # window in one persistent launch (profiles/r04_dfsph_loop_tail.txt).  Graph replay, W + K steps like the headline.
        if solver == "dfsph":
            result["reference_default_iterations"] = {}
            for arith in ("strict", "persistent"):
                sim, P = make_system(sphx, args.nx, solver, -1, -1, args.pbd_iters, arith=arith)
                if args.warmup > 0:
                    sim.step_n(args.warmup)
                wall, _ = timed_steps(torch, sim, args.steps)
                result["reference_default_iterations"][arith] = {"steps_per_s": args.steps / wall, "ms_per_step": wall * 1e3 / args.steps,
                                                                 "iterations_last_step": list(sim.iters()), "steps": args.steps, "warmup": args.warmup}
                note("adaptive DFSPH (%s): %.2f ms/step, iterations %s" % (arith, wall * 1e3 / args.steps, sim.iters()))
                sim.close()
        # Post-impact legs (ragged cells, wall contact, 40+ neighbours).  At 10 M particles the column hits the floor
        # around step 200; with the FIXED (1,4) iteration counts of config 5 the under-converged solve does not survive
        # that impact (densities and velocities run away within ~50 steps: tools/settle_probe.py), so the
        # 10 M leg runs the reference's own adaptive iteration control (thresholds 1e-3, at most 20 iterations); the
        # 1 M config keeps its fixed counts, which do survive.
        # r05: every arithmetic contract is timed here too -- the headline arithmetic is not only a free-fall number.
        result["steady_state"] = [settled_leg(sphx, torch, args.nx, "dfsph", -1, -1, args.settle_steps, args.steady_steps, arith=a)
                                  for a in ("strict", "tolerance", "persistent")]
        result["steady_state"] += [settled_leg(sphx, torch, 88, "dfsph", 1, 4, args.settle_steps, args.steady_steps, arith=a)
                                   for a in ("strict", "persistent")]
    if not args.no_extra_legs:
        legs = []
        # each in the parity contract (strict) and in the headline arithmetic (persistent; PBD keeps its own skin rows and runs as tolerance)
        for arith in ("strict", "persistent"):
            for nx, sv, steps in ((56, "wcsph", 200), (88, "dfsph", 100), (88, "pbd", 100)):      # BASELINE configs 2, 3, 4
                legs.append(small_leg(sphx, torch, nx, sv, 1, 4, 4, steps, 10, arith=arith))
            # the reference's own scene and defaults (20,736 particles; adaptive DFSPH, 20 PBD iterations), next to which
            # BASELINE.md quotes 4.4 / 23.0 / 11.3 ms per frame on a GTX 1070
            for sv, steps in (("wcsph", 300), ("dfsph", 100), ("pbd", 100)):
                legs.append(small_leg(sphx, torch, 24, sv, -1, -1, 20, steps, 10, arith=arith, landed=300 if sv == "wcsph" else 200))
        result["configs"] = legs
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, n)
    emit(result, real_stdout)


if __name__ == "__main__":
    main()
