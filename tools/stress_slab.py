"""Randomised stress of the native slab layer (loopback transport, one GPU): random container, slab count, solver, adaptive /
fixed DFSPH, overlap on/off, re-balancing cadence, splash state -- the gathered result must equal the single-domain oracle bit
for bit.  python tools/stress_slab.py [cases=100] [first_seed=0] [report file]
ARITH=1: the same cases under the TOLERANCE contract; the reference is then the single-device tolerance ENGINE (bit for bit as well: a
tolerance result is a function of a particle's row and inputs alone), both sides with fixed 96-entry rows (sphx_tuning.row_capacity) and PBD skin rows off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
import numpy as np
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
from oracle import oracle as O


ARITH = int(os.environ.get("ARITH", "0"))
if ARITH:
    os.environ["SPHX_NBR_CAP"] = "96"; os.environ["SPHX_PBD_SKIN"] = "0"


class _EngineRef:
    """the single-device engine behind the oracle's interface (tolerance runs)"""
    F_ID, F_VEL, F_POS, F_DENSITY, F_POS_LAST = sphx.F_ID, sphx.F_VEL, sphx.F_POS, sphx.F_DENSITY, sphx.F_POS_LAST


def run_case(seed):
    rng = np.random.default_rng(seed)
    nx = int(rng.choice([12, 16, 24]))
    P, fluid, boundary = sphx.scene(nx)
    solver = int(rng.integers(0, 3))
    ghost = 2 if solver == 2 else 1
    gx = P.cells[0]
    world = int(rng.integers(1, max(2, min(8, gx // (ghost + 1)) + 1)))
    P.solver = solver; P.dt = float(rng.choice([0.0005, 0.001])); P.pbd_iters = int(rng.integers(1, 5))
    adaptive = rng.random() < 0.4
    no_surface = rng.random() < 0.25
    if no_surface:
        P.surface_tension = 0.0; P.air_pressure = 0.0          # the stages without surface effects (separate add-delta-v / warm-start stages)
    if not adaptive:
        P.dfsph_fixed_div, P.dfsph_fixed_den = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    n = len(fluid)
    s = P.space[0]
    lo = 0.03 * s
    pos = rng.uniform(lo, rng.uniform(0.5, 0.93) * s, (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(lo, rng.uniform(0.15, 0.4) * s, n).astype(np.float32)
    vel = rng.normal(0, rng.choice([0.2, 0.6, 1.2]), (n, 3)).astype(np.float32)
    vel[:, 0] += np.where(pos[:, 0] < 0.5 * s, 1.0, -1.0).astype(np.float32) * np.float32(rng.choice([0.0, 1.5, 3.0]))
    flags = int(rng.choice([0, 0, sphx.SLAB_NO_OVERLAP]))
    steps = int(rng.integers(3, 9))
    desc = "seed %d nx %d world %d solver %d adaptive %s flags %d steps %d dt %g surface %s" % (seed, nx, world, solver, adaptive, flags, steps, P.dt, not no_surface)
    P.reserved[3] = ARITH
    Po = O.Params()
    for name, _ in P._fields_:
        setattr(Po, name, getattr(P, name))
    try:
        g = sphx.SlabGroup(P, pos, boundary, world, flags=flags, velocity=vel)
    except sphx.SphxError as e:
        return None if "too narrow" in str(e) else desc + " :: create failed: %s" % e
    o = sphx.System(P, pos, boundary, ctor_step=False) if ARITH else O.System(Po, pos, boundary, ctor_step=False)
    O_ = _EngineRef if ARITH else O
    try:
        if rng.random() < 0.6:
            g.set_rebalance(int(rng.integers(1, 4)), float(rng.choice([0.0, 0.05])))
        o.set(O_.F_VEL, vel[o.get(O_.F_ID)])
        for k in range(steps):
            try:
                g.step()
            except sphx.SphxError as e:      # a legitimate refusal (PBD only: a particle left the two ghost columns inside a step; DFSPH / WCSPH re-bin any displacement since r06)
                return None if "PBD: a particle moved more than one cell column" in str(e) else desc + " :: step %d failed: %s" % (k + 1, e)
            o.step()
            if solver == 2 and k == 0:
                o.set(O_.F_POS_LAST, (pos - np.float32(P.dt) * vel).astype(np.float32)[o.get(O_.F_ID)])
        ids, p, v, d = g.gather_all()
        order = np.argsort(o.get(O_.F_ID))
        for nm, a, b in (("pos", p, o.get(O_.F_POS)[order]), ("vel", v, o.get(O_.F_VEL)[order]), ("density", d, o.get(O_.F_DENSITY)[order])):
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                return desc + " :: %s differs in %d elements" % (nm, int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
        if solver == 1 and tuple(g.iters()) != tuple(o.iters()):
            return desc + " :: iterations %s vs %s" % (g.iters(), o.iters())
    finally:
        g.close(); o.close()
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    O.lib().oracle_set_threads(min(O.lib().oracle_max_threads(), 32))
    t0 = time.time(); failures = []
    for seed in range(first, first + cases):
        try:
            f = run_case(seed)
        except Exception as e:
            f = "seed %d :: exception %r" % (seed, e)
        if f:
            failures.append(f)
    summary = ("stress slab (tolerance arithmetic, reference = single-device tolerance engine): " if ARITH else "stress slab: ") + "%d cases (seeds %d..%d), %d failures, %.0f s" % (cases, first, first + cases - 1, len(failures), time.time() - t0)
    sys.stdout.flush()
    sys.stderr.write("\n" + "\n".join(["FAIL " + f for f in failures] + [summary]) + "\n")
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            f.write("\n".join(["FAIL " + x for x in failures] + [summary]) + "\n")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
