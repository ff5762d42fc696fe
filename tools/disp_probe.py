import sys, os
sys.path.insert(0, "/root/repo/cpp-fluid-particles_amd")
import numpy as np, sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
P, f, b = sphx.scene(56)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.reserved[3] = 1
s = sphx.System(P, f, b)
def bypid():
    ids = s.get(sphx.F_ID); p = s.get(sphx.F_POS); out = np.empty_like(p); out[ids] = p; return out.astype(np.float64)
R = P.radius; slack = P.cell_length - P.radius; lim = 0.45 * 0.95 * slack
print("R", R, "slack", slack, "limit", lim, "dt", P.dt)
for start in (1, 20):
    while True:
        st = s.persistent_stats()
        break
    p0 = bypid()
    for k in range(1, 7):
        s.step()
        d = bypid() - p0
        c = np.median(d, axis=0)
        r = np.linalg.norm(d - c, axis=1)
        v = s.get(sphx.F_VEL)
        print("after %d steps: |d-c| max %.3e  p99.9 %.3e p99 %.3e p90 %.3e median %.3e   frac > limit %.4f   |c| %.3e  vel spread %.3e" % (
            k, r.max(), np.percentile(r, 99.9), np.percentile(r, 99), np.percentile(r, 90), np.median(r), (r > lim).mean(), np.linalg.norm(c), np.ptp(v[:,1])))
    for _ in range(14): s.step()
