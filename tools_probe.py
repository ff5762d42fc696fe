import sys, time, os
sys.path.insert(0, "cpp-fluid-particles_amd")
import numpy as np, sphx
print("devices", sphx.device_count())
for nx, solver, name in ((56, sphx.WCSPH, "wcsph263k"), (88, sphx.DFSPH, "dfsph1m"), (88, sphx.PBD, "pbd1m")):
    P, f, b = sphx.scene(nx)
    P.solver = solver; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.pbd_iters = 4
    if solver == sphx.WCSPH: P.dt = 0.001
    t = time.time(); s = sphx.System(P, f, b); print(name, "n", s.n, "nb", s.nb, "create %.2fs" % (time.time() - t))
    s.step()
    s.step_n(10)
    ms = s.step_n(50)
    print(name, "ms/step %.3f steps/s %.1f" % (ms / 50, 50000.0 / ms))
    for nm, t in s.profile_step():
        print("   %-22s %8.3f ms" % (nm, t))
    s.close()
