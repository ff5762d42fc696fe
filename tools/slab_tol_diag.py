"""Diagnostic: where does a tolerance-arithmetic slab run part from the single-device tolerance engine?  For every quad mask of the
tolerance walks (sphx_tuning.quad_mask_tol) and both schedules: elements that differ bitwise after 1 and after 6 steps.
    python tools/slab_tol_diag.py [world=2] [solver=dfsph]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cpp-fluid-particles_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import sphx
import tuning_env; tuning_env.install(sphx)
import slab_worker

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
solver = sys.argv[2] if len(sys.argv) > 2 else "dfsph"
nx, seed = 12, 17
os.environ["SPHX_NBR_CAP"] = "96"; os.environ["SPHX_PBD_SKIN"] = "0"


def slabs(steps, flags, adaptive):
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive); P.reserved[3] = 1
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    g = sphx.SlabGroup(P, pos, boundary, world, flags=flags, velocity=vel)
    g.step(steps)
    out = g.gather_all(); g.close()
    return out[1], out[2], out[3]


def single(steps, adaptive):
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive); P.reserved[3] = 1
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    s = sphx.System(P, pos, boundary, ctor_step=False)
    s.set(sphx.F_VEL, vel[s.get(sphx.F_ID)])
    for _ in range(steps):
        s.step()
    o = np.argsort(s.get(sphx.F_ID))
    out = (s.get(sphx.F_POS)[o], s.get(sphx.F_VEL)[o], s.get(sphx.F_DENSITY)[o]); s.close()
    return out


for mask in ("0", "1", "2", "4", "8", "16", "7", "15"):
    os.environ["SPHX_QUAD_MASK_TOL"] = mask
    for adaptive in (False, True):
        for steps in (1, 6):
            ref = single(steps, adaptive)
            for flags in (0, 1):
                got = slabs(steps, flags, adaptive)
                diff = [int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))) for a, b in zip(got, ref)]
                print("quad_mask_tol %-3s adaptive %-5s steps %d flags %d: differing pos/vel/density elements %s" % (mask, adaptive, steps, flags, diff), flush=True)
