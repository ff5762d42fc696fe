for rep in 1 2; do for T in 0 2; do for cfg in wcsph263k dfsph1m pbd1m dfsph10m; do
  a=$(TOL=$T timeout 300 python tools/probe_step.py $cfg 2>/dev/null | grep "ms/step" | sed 's/.*ms\/step \([0-9.]*\).*/\1/')
  b=$(TOL=$T SPHX_LIB=$PWD/cpp-fluid-particles_amd/variants/old/libsphx.so timeout 300 python tools/probe_step.py $cfg 2>/dev/null | grep "ms/step" | sed 's/.*ms\/step \([0-9.]*\).*/\1/')
  echo "rep $rep TOL=$T $cfg: new $a | old $b"
done; done; done
