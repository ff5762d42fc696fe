import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import sphx
P, f, b = sphx.scene(88)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
s = sphx.System(P, f, b)
s.step(); s.step()
