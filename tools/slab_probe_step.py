"""per-kernel (hipEvent span) times of ONE slab-step of the native slab layer beside the plain engine's step, same scene and arithmetic:
where the slab layer's own overhead sits.     python tools/slab_probe_step.py [nx=190] [slabs=1] [arith=1]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cpp-fluid-particles_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import sphx
import tuning_env; tuning_env.install(sphx)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 190
slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
arith = int(sys.argv[3]) if len(sys.argv) > 3 else 1
P, f, b = sphx.scene(nx)
P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.reserved[3] = arith
s = sphx.System(P, f, b); s.step_n(6)
t0 = time.perf_counter(); s.step_n(10); torch.cuda.synchronize(); plain_ms = (time.perf_counter() - t0) * 100
plain = dict(s.profile_step()); s.close()
g = sphx.SlabGroup(P, f, b, slabs); g.step(7)
t0 = time.perf_counter(); g.step(10); torch.cuda.synchronize(); slab_ms = (time.perf_counter() - t0) * 100
sphx.kernel_timer(True, "")
K = 4
g.step(K)
spans = sphx.kernel_timer_collect(256); sphx.kernel_timer(False); g.close()
print("nx %d, arithmetic %d: plain engine %.3f ms/step (graph replay), %d slab(s) %.3f ms/step" % (nx, arith, plain_ms, slabs, slab_ms))
print("%-26s %10s %10s %8s" % ("span", "plain ms", "slab ms", "launches/step"))
names = list(plain) + [k for k in spans if k not in plain]
tp = ts = 0.0
for nm in names:
    a = plain.get(nm, 0.0); bms, cnt = spans.get(nm, (0.0, 0)); bms /= K
    tp += a; ts += bms
    print("%-26s %10.3f %10.3f %8.1f" % (nm, a, bms, cnt / float(K)))
print("%-26s %10.3f %10.3f   (spans only: the slab layer's own exchange kernels and copies are not inside spans)" % ("sum of spans", tp, ts))
