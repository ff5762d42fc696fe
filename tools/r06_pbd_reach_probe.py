"""why three PBD stress seeds differ: farthest x-travel per step (columns) in the oracle run of the seed, against the slab run's first differing step"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, sphx
from oracle import oracle as O
for seed in [int(a) for a in sys.argv[1:]] or [5038, 5012]:
    rng = np.random.default_rng(seed)
    nx = int(rng.choice([12, 16, 24])); P, fluid, boundary = sphx.scene(nx)
    solver = int(rng.integers(0, 3)); ghost = 2 if solver == 2 else 1; gx = P.cells[0]
    world = int(rng.integers(1, max(2, min(8, gx // (ghost + 1)) + 1)))
    P.solver = solver; P.dt = float(rng.choice([0.0005, 0.001])); P.pbd_iters = int(rng.integers(1, 5))
    adaptive = rng.random() < 0.4; no_surface = rng.random() < 0.25
    if no_surface: P.surface_tension = 0.0; P.air_pressure = 0.0
    if not adaptive: P.dfsph_fixed_div, P.dfsph_fixed_den = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    n = len(fluid); s = P.space[0]; lo = 0.03 * s
    pos = rng.uniform(lo, rng.uniform(0.5, 0.93) * s, (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(lo, rng.uniform(0.15, 0.4) * s, n).astype(np.float32)
    vel = rng.normal(0, rng.choice([0.2, 0.6, 1.2]), (n, 3)).astype(np.float32)
    vel[:, 0] += np.where(pos[:, 0] < 0.5 * s, 1.0, -1.0).astype(np.float32) * np.float32(rng.choice([0.0, 1.5, 3.0]))
    flags = int(rng.choice([0, 0, sphx.SLAB_NO_OVERLAP])); steps = int(rng.integers(3, 9))
    Po = O.Params()
    for name, _ in P._fields_: setattr(Po, name, getattr(P, name))
    g = sphx.SlabGroup(P, pos, boundary, world, flags=flags, velocity=vel)
    o = O.System(Po, pos, boundary, ctor_step=False)
    if rng.random() < 0.6: g.set_rebalance(int(rng.integers(1, 4)), float(rng.choice([0.0, 0.05])))
    o.set(O.F_VEL, vel[o.get(O.F_ID)])
    print("seed %d nx %d world %d solver %d steps %d flags %d cuts %s" % (seed, nx, world, solver, steps, flags, [g.info(i)[:2] for i in range(world)]))
    prev = pos.copy()
    for k in range(steps):
        g.step(); o.step()
        if solver == 2 and k == 0: o.set(O.F_POS_LAST, (pos - np.float32(P.dt) * vel).astype(np.float32)[o.get(O.F_ID)])
        order = np.argsort(o.get(O.F_ID)); op = o.get(O.F_POS)[order]
        travel = np.abs(op[:, 0] - prev[:, 0]) / P.cell_length
        ids, p, v, d = g.gather_all()
        nd = int(np.count_nonzero((p.view(np.uint32) != op.view(np.uint32)).any(axis=1)))
        print("   step %d: farthest x-travel %.2f columns (%d particles > 1 column); slab positions differing from the oracle: %d; cuts %s" % (k + 1, travel.max(), int((travel > 1).sum()), nd, [g.info(i)[:2] for i in range(world)]))
        prev = op.copy()
    g.close(); o.close()
