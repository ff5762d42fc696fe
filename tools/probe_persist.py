"""Step time of the tolerance engine with and without persistent rows (reserved[3] = 1 | 2), per-kernel breakdown and how often
the rows were rebuilt.   python tools/probe_persist.py [dfsph10m dfsph1m wcsph263k]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
which = sys.argv[1:] or ["dfsph10m"]
cfg = {"wcsph263k": (56, sphx.WCSPH), "dfsph1m": (88, sphx.DFSPH), "dfsph10m": (190, sphx.DFSPH), "wcsph10m": (190, sphx.WCSPH)}
for name in which:
    nx, solver = cfg[name]
    for mode in (1, 2):
        P, f, b = sphx.scene(nx)
        P.solver = solver; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
        P.reserved[3] = mode
        if solver == sphx.WCSPH: P.dt = 0.001
        s = sphx.System(P, f, b)
        s.step()
        s.step_n(5)
        for rep in range(4):
            ms = s.step_n(20)
            print(name, "arith", mode, "steps %3d-%3d  ms/step %.3f  steps/s %.1f  (in use, row builds, steps) = %s" % (7 + 20 * rep, 26 + 20 * rep, ms / 20, 20000.0 / ms, s.persistent_stats()), flush=True)
        for nm, t in s.profile_step():
            print("   %-22s %8.3f ms" % (nm, t))
        s.close()
