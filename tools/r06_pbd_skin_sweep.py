"""PBD skin rows after the landing: ms/step and in-step row rebuilds per step for fixed skin widths (in units of R) -- the reference scene under
its default 20 iterations and BASELINE config 4 (1,022,208 particles, 4 iterations).   python tools/r06_pbd_skin_sweep.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
for name, nx, iters, settle in (("reference scene, PBD(20)", 24, 20, 250), ("config 4, PBD(4)", 88, 4, 300)):
    for skin in (0.0, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4):
        sphx.set_tuning(pbd_skin=skin, pbd_skin_fixed=1)
        P, f, b = sphx.scene(nx)
        P.solver = sphx.PBD; P.dt = 0.002; P.pbd_iters = iters
        s = sphx.System(P, f, b)
        s.step_n(10)
        ff = min(s.step_n(20) / 20 for _ in range(2))
        s.step_n(settle)
        r0 = s.rows_stale(); q0 = s.rows_partial()
        ms = min(s.step_n(50) / 50 for _ in range(2))
        r1 = s.rows_stale(); q1 = s.rows_partial()
        tot, mx, hist = s.row_stats()
        print("%s, skin %.2f R: free fall %.3f ms/step | landed %.3f ms/step, %.2f whole and %.2f row-wise rebuilds per step, rows: mean %.1f longest %d, capacity %d"
              % (name, skin, ff, ms, (r1 - r0) / 100.0, (q1 - q0) / 100.0, tot / s.n, mx, sphx.row_capacity(s)), flush=True)
        s.close()
sphx.set_tuning()
