import sys
sys.path.insert(0, "cpp-fluid-particles_amd")
import sphx
for R in (0.04, 0.013, 0.5):
    print(R, sphx.fastmath_selftest(R, 1 << 26))
