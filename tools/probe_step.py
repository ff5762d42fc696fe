import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
which = sys.argv[1:] or ["wcsph263k", "dfsph1m", "pbd1m"]
cfg = {"wcsph263k": (56, sphx.WCSPH), "dfsph1m": (88, sphx.DFSPH), "pbd1m": (88, sphx.PBD), "dfsph10m": (190, sphx.DFSPH)}
for name in which:
    nx, solver = cfg[name]
    P, f, b = sphx.scene(nx)
    P.solver = solver; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.pbd_iters = 4
    P.reserved[0] = int(os.environ.get("FLAGS", "0")); P.reserved[3] = int(os.environ.get("TOL", "0"))
    if solver == sphx.WCSPH: P.dt = 0.001
    t = time.time(); s = sphx.System(P, f, b); print(name, "n", s.n, "nb", s.nb, "create %.2fs" % (time.time() - t), flush=True)
    s.step()
    s.step_n(10)
    ms = s.step_n(50)
    print(name, "flags", P.reserved[0], "ms/step %.3f steps/s %.1f" % (ms / 50, 50000.0 / ms))
    for nm, t in s.profile_step():
        print("   %-22s %8.3f ms" % (nm, t))
    s.close()
