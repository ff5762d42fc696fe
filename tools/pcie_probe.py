"""PCIe-inclusive rate of the bench workload: one SPHSystem::step() plus a host read-back of the
positions through sphx_get (what a caller without GPU-side consumers would do every step).
Usage: python tools/pcie_probe.py [nx]   (default 190 = the 10.3 M-particle bench scene)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np
import sphx

import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
P, f, b = sphx.scene(nx)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
t0 = time.time(); s = sphx.System(P, f, b); t_create = time.time() - t0
s.step_n(5)
ms = s.step_n(20) / 20
pos = s.get(sphx.F_POS)              # first read-back (allocates the host array)
t0 = time.perf_counter()
for _ in range(5):
    pos = s.get(sphx.F_POS)
t_get = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(10):
    s.step(); pos = s.get(sphx.F_POS)
t_both = (time.perf_counter() - t0) / 10
print("n=%d create(upload+ctor step) %.2f s | step %.3f ms (%.1f steps/s) | read-back of pos (%.1f MB) %.3f ms = %.1f GB/s | "
      "step + read-back %.3f ms (%.1f steps/s PCIe-inclusive)"
      % (s.n, t_create, ms, 1e3 / ms, pos.nbytes / 1e6, t_get * 1e3, pos.nbytes / t_get / 1e9, t_both * 1e3, 1.0 / t_both))
s.close()
