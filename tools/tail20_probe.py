import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
P, f, b = sphx.scene(24); P.solver = sphx.DFSPH
s = sphx.System(P, f, b); s.step(); s.step_n(149)
print(os.environ.get("SPHX_LIB", "default")[-24:], os.environ.get("SPHX_DFSPH_NO_TAIL", "-"), "ms/step (20 divergence iterations per step):", ["%.3f" % (s.step_n(50) / 50) for r in range(3)], flush=True)
s.close()
