"""max |v| and the farthest x-travel per step (in cell columns) of the plain engine on the config-5 scene under the reference's adaptive
control: what a slab exchange has to reach.  python tools/r06_vmax_probe.py [nx=190] [steps=330]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp-fluid-particles_amd"))
import numpy as np, sphx
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 330
P, fluid, boundary = sphx.scene(nx)
P.solver = sphx.DFSPH
s = sphx.System(P, fluid, boundary)
done = 1
worst = 0.0
while done < steps:
    s.step_n(8); done += 8
    v = s.get(sphx.F_VEL)
    sp = np.sqrt((v.astype(np.float64) ** 2).sum(1))
    cols = np.abs(v[:, 0]).max() * P.dt / P.cell_length
    worst = max(worst, cols)
    if done % 32 == 1 or cols > 1.0:
        print("step %4d  |v| max %9.2f  p99.99 %8.2f  farthest x-travel %6.2f columns  (> 1 column: %d particles)  iters %s" % (
            done, sp.max(), np.quantile(sp, 0.9999), cols, int(np.count_nonzero(np.abs(v[:, 0]) * P.dt > P.cell_length)), s.iters()), flush=True)
print("worst x-travel per step: %.2f columns" % worst)
