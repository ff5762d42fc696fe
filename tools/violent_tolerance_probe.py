"""What do the tolerance engines do in the first steps from a post-impact state of the strict engine?  Prints, per step, the deviation
of the tolerance / persistent engines AND of the one-ulp-perturbed strict control from the ORACLE (max and quantiles), the numbers
tests/test_gpu_violent.py::test_tolerance_engines_from_post_impact_states asserts on.
    python tools/violent_tolerance_probe.py [nx=88] [settle=300] [fixed=1 | adaptive=0]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cpp-fluid-particles_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
from oracle import oracle
import test_gpu_violent as T

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 88
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fixed = (1, 4) if (len(sys.argv) <= 3 or sys.argv[3] == "1") else (-1, -1)
oracle.lib().oracle_set_threads(oracle.lib().oracle_max_threads())
P, boundary, st = T.engine_state(sphx, nx, fixed, settle)
print("state: nx %d, %s, step %d: rho max %.3f, |v| max %.2f, longest row %d, iterations %s" % (nx, fixed, settle, st["rho_max"], np.abs(st["vel"]).max(), st["rows"], st["iters"]))
o = T.restart(oracle, P, boundary, st)
runs = {"tolerance": T.restart(sphx, P, boundary, st, arith=1), "persistent": T.restart(sphx, P, boundary, st, arith=2)}
rng = np.random.default_rng(settle)
pos1 = np.where(rng.random(st["pos"].shape) < 0.5, np.nextafter(st["pos"], np.float32(2)), st["pos"]).astype(np.float32)
runs["one-ulp control (strict)"] = T.restart(sphx, P, boundary, st, pos=pos1)
for step in range(4):
    o.step()
    rp, rr, rv = (T._by_id(oracle, o, f).astype(np.float64) for f in (oracle.F_POS, oracle.F_DENSITY, oracle.F_VEL))
    print("step +%d: oracle iterations %s, |v|max %.1f, rho max %.2f" % (step + 1, o.iters(), np.abs(rv).max(), rr.max()))
    for name, g in runs.items():
        g.step()
        p = T._by_id(sphx, g, sphx.F_POS).astype(np.float64); r = T._by_id(sphx, g, sphx.F_DENSITY).astype(np.float64)
        dp = np.abs(p - rp).max(axis=1) / P.space[0]; dr = np.abs(r - rr) / P.rho0
        q = lambda a: "max %.2e  p99.99 %.2e  p99.9 %.2e  p99 %.2e  median %.2e  over 1e-5: %d" % (a.max(), np.quantile(a, 0.9999), np.quantile(a, 0.999), np.quantile(a, 0.99), np.median(a), int((a > 1e-5).sum()))
        cells = int(np.count_nonzero(T._by_id(sphx, g, sphx.F_CELL) != T._by_id(oracle, o, oracle.F_CELL)))
        print("   %-26s iters %-8s pos/domain: %s" % (name, g.iters(), q(dp)))
        print("   %-26s cells differ %-6d rho/rho0:   %s" % ("", cells, q(dr)))
print("persistent stats", runs["persistent"].persistent_stats())
