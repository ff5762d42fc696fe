"""the engine's opt-in compact-brick schedule (sphx_tuning.brick = 1, tolerance arithmetic, rows every step) against the quad walks at
10.3 M particles: ms per step and one profiled step per kernel.   python tools/r06_brick_engine_probe.py [nx=190]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
for brick in (0, 1):
    sphx.set_tuning(brick=brick)
    P, f, b = sphx.scene(nx)
    P.solver = sphx.DFSPH; P.dfsph_fixed_div, P.dfsph_fixed_den = 1, 4; P.reserved[3] = 1
    s = sphx.System(P, f, b)
    s.step_n(5)
    ms = min(s.step_n(20) / 20 for _ in range(2))
    prof = s.profile_step()
    agg = {}
    for nm, t in prof:
        a = agg.setdefault(nm, [0.0, 0]); a[0] += t; a[1] += 1
    print("brick=%d: %.3f ms/step; profiled step: %s" % (brick, ms, ", ".join("%s %.3f (x%d)" % (nm, t, c) for nm, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]))), flush=True)
    s.close()
sphx.set_tuning()
