for m in default 255; do for a in 0 1; do
  if [ $m = default ]; then unset SPHX_QUAD_MASK SPHX_QUAD_MASK_TOL; else export SPHX_QUAD_MASK=$m SPHX_QUAD_MASK_TOL=$m; fi
  echo "== quad mask $m arith $a"; TOL=$a python tools/small_probe.py 2>/dev/null | grep -v "amdgpu\|^PBD"
done; done
unset SPHX_QUAD_MASK SPHX_QUAD_MASK_TOL
for m in default 255; do for a in 0 1; do
  if [ $m = default ]; then unset SPHX_QUAD_MASK SPHX_QUAD_MASK_TOL; else export SPHX_QUAD_MASK=$m SPHX_QUAD_MASK_TOL=$m; fi
  echo "== quad mask $m arith $a"; TOL=$a python tools/probe_step.py wcsph263k pbd1m 2>/dev/null | grep "ms/step"
done; done
