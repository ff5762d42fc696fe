"""Regenerates tests/golden/dambreak_nx8.npz from the CPU oracle (which is itself pinned to the
reference through the SURVEY §8(c) known answers, see tests/test_cpu_oracle.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

STEPS = 10
out = {"steps": np.int32(STEPS)}
for key, solver in (("wcsph", O.WCSPH), ("dfsph", O.DFSPH), ("pbd", O.PBD)):
    P, fluid, boundary = O.scene(8)
    P.solver = solver
    P.pbd_iters = 4
    s = O.System(P, fluid, boundary)
    for _ in range(STEPS):
        s.step()
    out[key + "_pos"] = s.get(O.F_POS)
    out[key + "_density"] = s.get(O.F_DENSITY)
    out[key + "_cell"] = s.get(O.F_CELL)
    out[key + "_iters"] = np.array(s.iters(), np.int32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dambreak_nx8.npz"), **out)
print("written", {k: getattr(v, "shape", None) for k, v in out.items()})
