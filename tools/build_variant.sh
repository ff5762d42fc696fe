#!/bin/bash
# Builds a VARIANT of libsphx.so with extra compile flags into cpp-fluid-particles_amd/variants/<name>/libsphx.so (git-ignored; it
# travels to the GPU box) for A/B measurements: select it with SPHX_LIB.     tools/build_variant.sh nt -DSPHX_NT_ROWS=1
set -eu
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); PKG=$R/cpp-fluid-particles_amd; OUT=$PKG/variants/$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function -I$R/include -I$PKG/csrc"
for f in runtime system wcsph dfsph pbd capi slab obstacles; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $PKG/csrc/$f.hip -o $OUT/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsphx.so $OUT/*.o -ldl
rm -f $OUT/*.o
echo $OUT/libsphx.so
