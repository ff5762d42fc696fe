"""Which post-impact states do tests/test_gpu_violent.py hand over?  Prints, every 20 steps, what the strict engine's dam break
looks like (largest density, speed, row length, iteration counts, run-away or not) for a list of scene sizes and controls:
    python tools/violent_probe.py 88:a 112:f 128:f 144:f [steps=400]
(`a` = the reference's adaptive control, `f` = fixed (1,4))"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
import numpy as np
import torch
import sphx

import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
cases = [a for a in sys.argv[1:] if ":" in a]
total = next((int(a) for a in sys.argv[1:] if ":" not in a), 400)
for case in cases:
    nx, kind = case.split(":"); nx = int(nx)
    P, fluid, boundary = sphx.scene(nx)
    P.solver = sphx.DFSPH
    P.dfsph_fixed_div, P.dfsph_fixed_den = (1, 4) if kind == "f" else (-1, -1)
    s = sphx.System(P, fluid, boundary)
    done = 1
    print("== nx %d (%d particles), %s" % (nx, s.n, "fixed (1,4)" if kind == "f" else "adaptive"), flush=True)
    while done < total:
        t0 = time.perf_counter(); s.step_n(20); dt = (time.perf_counter() - t0) / 20; done += 20
        den = s.get(sphx.F_DENSITY); vel = s.get(sphx.F_VEL); pos = s.get(sphx.F_POS)
        tot, mx, hist = s.row_stats()
        print("step %4d  %.2f ms/step  rows max %d  rho max %.3f  |v|max %.2f  finite %s  iters %s" % (
            done, dt * 1e3, mx, den.max(), np.abs(vel).max(), bool(np.isfinite(pos).all() and np.isfinite(vel).all()), s.iters()), flush=True)
        if dt > 0.25 or not np.isfinite(vel).all():
            print("run-away: stopping"); break
    s.close()
