"""PBD skin rows: step time and how often the rows go stale, for a few skin widths (env SPHX_PBD_SKIN, in units of R),
in free fall (steps 1-60) and after the impact (steps 300-360).  usage: python tools/pbd_skin_probe.py [nx=88]"""
import os, subprocess, sys
nx = sys.argv[1] if len(sys.argv) > 1 else "88"
code = r'''
import sys, os, time
sys.path.insert(0, os.path.join(%r, "cpp-fluid-particles_amd"))
import sphx
import sys as _sys, os as _os; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)
P, f, b = sphx.scene(%s); P.solver = sphx.PBD; P.pbd_iters = 4
s = sphx.System(P, f, b); s.step()
def leg(n):
    before = s.rows_stale(); t0 = time.perf_counter()
    s.step_n(n)
    return (time.perf_counter() - t0) * 1e3 / n, s.rows_stale() - before
a = leg(60); s.step_n(240); b2 = leg(60)
tot, mx, hist = s.row_stats()
print("skin %%s R: free fall %%.3f ms/step, %%d in-step rebuilds in 60 steps | post-impact %%.3f ms/step, %%d rebuilds in 60 steps | row mean %%.1f max %%d" %% (os.environ.get("SPHX_PBD_SKIN", "0.1"), a[0], a[1], b2[0], b2[1], tot / s.n, mx))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), nx)
for skin in ("0", "0.03", "0.05", "0.1", "0.2"):
    env = dict(os.environ, SPHX_PBD_SKIN=skin, SPHX_PBD_SKIN_FIXED="1")     # fixed skins: the controller is what this probe informs
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    hits = [l for l in out.stdout.splitlines() if l.startswith("skin")]
    print(hits[-1] if hits else out.stderr[-300:])
