"""where do quad-per-particle walks in EVERY sweep stop paying off?  ms/step by scene size for the default masks and for mask 255, all three
solvers (reference defaults), strict and tolerance arithmetic, free-fall window.     python tools/quad_size_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cpp-fluid-particles_amd"), os.path.join(ROOT, "tests")]
import sphx
import tuning_env; tuning_env.install(sphx)
for nx in (24, 32, 40, 48, 56):
    for name, solver, dt in (("wcsph", sphx.WCSPH, 0.001), ("dfsph", sphx.DFSPH, 0.002), ("pbd20", sphx.PBD, 0.002)):
        row = []
        for arith in (0, 1):
            for mask in (None, "255"):
                for k in ("SPHX_QUAD_MASK", "SPHX_QUAD_MASK_TOL"):
                    os.environ.pop(k, None)
                    if mask:
                        os.environ[k] = mask
                P, f, b = sphx.scene(nx)
                P.solver = solver; P.dt = dt; P.reserved[3] = arith
                s = sphx.System(P, f, b)
                s.step_n(10)
                row.append(s.step_n(40) / 40)
                n = s.n
                s.close()
        print("nx %2d n %7d %-6s strict default %.3f all-quads %.3f | tolerance default %.3f all-quads %.3f" % (nx, n, name, row[0], row[1], row[2], row[3]), flush=True)
