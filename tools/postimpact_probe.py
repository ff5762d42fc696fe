"""per-kernel times of the 10 M adaptive DFSPH scene after its impact (step 275), for the engine flags given in FLAGS
   FLAGS=16 python tools/postimpact_probe.py [nx=190] [steps=275]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
import numpy as np
import torch
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
total = int(sys.argv[2]) if len(sys.argv) > 2 else 275
P, fluid, boundary = sphx.scene(nx)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = -1; P.dfsph_fixed_den = -1
P.reserved[0] = int(os.environ.get("FLAGS", "0")); P.reserved[3] = int(os.environ.get("TOL", "0"))
s = sphx.System(P, fluid, boundary)
s.step_n(total)
t0 = time.perf_counter(); s.step_n(10); dt = (time.perf_counter() - t0) / 10
tot, mx, hist = s.row_stats()
print("flags %d: step %d: %.2f ms/step, iterations %s, neighbours mean %.1f max %d, rows over 32: %d, over 64: %d" % (
    P.reserved[0], total + 10, dt * 1e3, s.iters(), tot / s.n, mx, int(hist[33:].sum()), int(hist[65:].sum())), flush=True)
for nm, t in s.profile_step():
    if t > 0.3:
        print("   %-24s %9.3f ms" % (nm, t))
