"""reference scene (20,736 particles), PBD(20): per-kernel times of one profiled step in free fall and after the landing, beside the batch time per step.
   python tools/r06_pbd_landed_probe.py [nx]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import sphx, tuning_env
tuning_env.install(sphx)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for arith in (0, 1):
    P, f, b = sphx.scene(nx)
    P.solver = sphx.PBD; P.dt = 0.002; P.reserved[3] = arith
    if os.environ.get("ITERS"): P.pbd_iters = int(os.environ["ITERS"])
    s = sphx.System(P, f, b)
    s.step_n(10)
    for settle in (0, 200, 200):
        if settle: s.step_n(settle)
        ms = min(s.step_n(50) / 50 for _ in range(2))
        prof = s.profile_step()
        agg = {}
        for nm, t in prof:
            a = agg.setdefault(nm, [0.0, 0]); a[0] += t; a[1] += 1
        print("arith %d, after %d more steps: %.3f ms/step in batches; profiled step: %d launches, %.3f ms of kernels (cap %d, longest row %d)"
              % (arith, settle, ms, len(prof), sum(t for _, t in prof), sphx.row_capacity(s), s.row_stats()[1]))
        print("   " + ", ".join("%s %.3f (x%d)" % (nm, t, c) for nm, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])), flush=True)
    s.close()
