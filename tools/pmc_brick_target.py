import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
P, f, b = sphx.scene(int(os.environ.get("NX", "190")))
P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.reserved[3] = int(os.environ.get("TOL", "1"))
s = sphx.System(P, f, b)
s.step(); s.step()
