"""reference scene (20,736 particles): N steps of one solver behind a settle phase, for a rocprofv3 --kernel-trace --stats run.
   python tools/r06_refscene_trace.py wcsph|dfsph|pbd settle steps"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sphx, tuning_env
tuning_env.install(sphx)
name, settle, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
P, f, b = sphx.scene(int(os.environ.get("NX", "24")))
P.solver = {"wcsph": sphx.WCSPH, "dfsph": sphx.DFSPH, "pbd": sphx.PBD}[name]; P.dt = 0.001 if name == "wcsph" else 0.002
P.reserved[3] = int(os.environ.get("TOL", "0"))
s = sphx.System(P, f, b)
s.step_n(10 + settle)
ms = s.step_n(steps)
print("%s after %d steps: %.3f ms/step over %d steps" % (name, 10 + settle, ms / steps, steps))
s.close()
