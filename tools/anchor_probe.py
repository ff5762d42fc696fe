"""Engine (strict contract: D1 fp64 x^7 chain, D3 Jacobi XSPH) against the statistics tests/golden/refsrc_anchors.json holds of the
REFERENCE SOURCES (libm powf, in-place XSPH): relative deviation of rho_mean / rho_min / rho_max / mean_y / vmax at every anchor
state of WCSPH (dt = 0.001, steps 0-300) and PBD(20) (dt = 0.002, steps 0-120).  The numbers behind
tests/test_gpu_parity.py::test_reference_source_anchor_statistics_wcsph_pbd_on_gpu."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
V = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "refsrc_anchors.json")))["variants"]["float_fabs"]
def stats(s):
    rho = s.get(sphx.F_DENSITY).astype(np.float64); pos = s.get(sphx.F_POS).astype(np.float64); vel = s.get(sphx.F_VEL).astype(np.float64)
    return {"rho_mean": rho.mean(), "rho_min": rho.min(), "rho_max": rho.max(), "mean_y": pos[:, 1].mean(), "vmax": np.sqrt((vel * vel).sum(1)).max()}
for name, solver in (("wcsph", sphx.WCSPH), ("pbd", sphx.PBD)):
    A = V[name]
    P, f, b = sphx.scene(24)
    P.solver = solver; P.dt = A["dt"]
    s = sphx.System(P, f, b)
    at = 0
    for st in A["states"]:
        while at < st["step"]:
            s.step(); at += 1
        got = stats(s)
        dev = {k: abs(got[k] - st[k]) / max(abs(st[k]), 1e-30) for k in got}
        print(name, "step %3d" % st["step"], "  ".join("%s %.2e" % (k, dev[k]) for k in dev), " rho_max %.6f" % got["rho_max"], flush=True)
    s.close()
