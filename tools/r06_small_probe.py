"""per-kernel times of one profiled step of BASELINE configs 2 (263,424 WCSPH) and 4 (1,022,208 PBD(4)), strict and headline arithmetic,
beside the batch time per step: where a small step's time goes.   python tools/r06_small_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
for name, nx, solver, dt in (("config 2", 56, sphx.WCSPH, 0.001), ("config 4", 88, sphx.PBD, 0.002)):
    for arith in (0, 2):
        P, f, b = sphx.scene(nx)
        P.solver = solver; P.dt = dt; P.pbd_iters = 4; P.reserved[3] = arith
        s = sphx.System(P, f, b)
        s.step_n(10)
        ms = min(s.step_n(100) / 100 for _ in range(3))
        prof = s.profile_step()
        tot = sum(t for _, t in prof)
        print("%s (%d particles) arith %d: %.3f ms/step in batches; one profiled step: %.3f ms of kernels in %d launches" % (name, len(f), arith, ms, tot, len(prof)))
        agg = {}
        for nm, t in prof:
            a = agg.setdefault(nm, [0.0, 0]); a[0] += t; a[1] += 1
        print("   " + ", ".join("%s %.3f (x%d)" % (nm, t, c) for nm, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])))
        s.close()
