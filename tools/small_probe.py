"""reference scene (20,736 particles), default solver settings, step_n batches through the landing: ms/step and row capacity"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
for name, solver, dt in (("wcsph", sphx.WCSPH, 0.001), ("dfsph", sphx.DFSPH, 0.002), ("pbd20", sphx.PBD, 0.002)):
    P, f, b = sphx.scene(24)
    P.solver = solver; P.dt = dt; P.reserved[3] = int(os.environ.get("TOL", "0"))
    s = sphx.System(P, f, b)
    s.step_n(10)
    out = []
    for batch in range(3):
        ms = s.step_n(100)
        out.append("%.3f ms/step (cap %d, longest row %d)" % (ms / 100, sphx.row_capacity(s), s.row_stats()[1]))
    print(name, " | ".join(out), flush=True)
    s.close()
