"""how does the headline workload behave after impact?  ms/step and neighbour statistics per 25 steps"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
import numpy as np
import torch
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
total = int(sys.argv[2]) if len(sys.argv) > 2 else 450
div, den = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 4)
P, fluid, boundary = sphx.scene(nx)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = div; P.dfsph_fixed_den = den
P.reserved[3] = int(os.environ.get("TOL", "0"))      # 0 strict, 1 tolerance, 2 tolerance + persistent rows
s = sphx.System(P, fluid, boundary)
done = 0
while done < total:
    t0 = time.perf_counter(); s.step_n(25); dt = (time.perf_counter() - t0) / 25; done += 25
    tot, mx, hist = s.row_stats()
    den_ = s.get(sphx.F_DENSITY); vel = s.get(sphx.F_VEL); pos = s.get(sphx.F_POS)
    print("step %4d  %.2f ms/step  nbrs mean %.1f max %d  rho mean %.3f max %.3f  |v|max %.2f  ymin %.3f finite %s  iters %s  persistent %s" % (
        done, dt * 1e3, tot / s.n, mx, den_.mean(), den_.max(), np.abs(vel).max(), pos[:, 1].min(), np.isfinite(pos).all(), s.iters(), s.persistent_stats()), flush=True)
    if dt > 0.2:
        print("pathological: stopping"); break
