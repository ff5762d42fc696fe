"""Adaptive DFSPH with the loop tail (one persistent launch per solver loop) against gated launches (SPHX_DFSPH_NO_TAIL=1):
states and iteration counts must be identical (strict arithmetic: bit for bit), step times side by side.
   python tools/tail_probe.py [nx=24] [steps=300] [arith=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
arith = int(sys.argv[3]) if len(sys.argv) > 3 else 0

def run(tail, timed):
    if tail: os.environ.pop("SPHX_DFSPH_NO_TAIL", None)
    else: os.environ["SPHX_DFSPH_NO_TAIL"] = "1"
    P, f, b = sphx.scene(nx)
    P.solver = sphx.DFSPH; P.reserved[3] = arith
    s = sphx.System(P, f, b)
    its = []; ms = []
    if timed:
        s.step()
        for w in range(steps // 50):
            ms.append(s.step_n(50) / 50); its.append(s.iters())
    else:
        for k in range(steps):
            s.step(); its.append(s.iters())
    if timed: print("   per-kernel (one profiled step):", ", ".join("%s %.3f" % (nm, t) for nm, t in s.profile_step()))
    out = (s.get(sphx.F_POS).copy(), s.get(sphx.F_VEL).copy(), s.get(sphx.F_DENSITY).copy(), its, ms)
    s.close()
    return out
a = run(True, False); b = run(False, False)
print("nx %d arith %d: %d steps; iteration counts equal: %s; max (div, den) = (%d, %d); pos/vel/density bitwise equal: %s %s %s" % (
    nx, arith, steps, a[3] == b[3], max(x[0] for x in a[3]), max(x[1] for x in a[3]),
    np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)),
    np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))), flush=True)
print("mean (div, den) iterations per window of 50 steps:", [(round(float(np.mean([x[0] for x in a[3][w:w + 50]])), 1), round(float(np.mean([x[1] for x in a[3][w:w + 50]])), 1)) for w in range(0, steps, 50)])
if a[3] != b[3]:
    for k, (x, y) in enumerate(zip(a[3], b[3])):
        if x != y: print("  first difference at step", k, x, y); break
ta = run(True, True); tb = run(False, True)
print("ms/step per window of 50 steps   tail :", " ".join("%.3f" % x for x in ta[4]))
print("                                 gated:", " ".join("%.3f" % x for x in tb[4]))
print("iterations at window ends        tail :", ta[3]); print("                                 gated:", tb[3])
