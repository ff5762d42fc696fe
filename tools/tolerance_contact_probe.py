"""Per-step deviation of the tolerance-arithmetic engine from the strict oracle, started from an identical
pre-impact state of the reference scene (GPU box): scaled metric and element-wise relative metric."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
from oracle import oracle as O
from test_gpu_tolerance import restart_pair, deviations
for solver, dt, k0, horizon in ((0, 0.001, 125, 60), (1, 0.002, 55, 50), (2, 0.002, 50, 50)):
    g, o, P = restart_pair(sphx, O, solver, dt, k0)
    for s in range(1, horizon + 1):
        g.step(); o.step()
        d = deviations(sphx, O, g, o, P)
        if s % 5 == 0 or s == 1:
            print("solver %d step %d+%d:" % (solver, k0, s), " ".join("%s=%.2e" % kv for kv in d.items()), "rho_max %.3f" % o.get(O.F_DENSITY).max(), flush=True)
