# one GPU call: the 16-lanes-per-particle row builder of small scenes -- parity first, then what it buys and up to which size
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tolerance.py -x -q > gpurun_out/r06_group_parity.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_group_parity.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_slab.py -x -q > gpurun_out/r06_group_slab.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_group_slab.log | tail -3
echo "--- reference scene, step_n batches through the landing (group builder | lane-per-particle builder)"
timeout 300 python tools/small_probe.py 2>&1 | grep -v "^PBD"
SPHX_GROUP_BUILD_MAX=-1 timeout 300 python tools/small_probe.py 2>&1 | grep -v "^PBD"
echo "--- the same in the headline arithmetic (TOL=2)"
TOL=2 timeout 300 python tools/small_probe.py 2>&1 | grep -v "^PBD"
echo "--- crossover: WCSPH ms/step by scene size, group builder forced on / off"
for nx in 16 24 28 32 36 40; do
  a=$(NX=$nx SPHX_GROUP_BUILD_MAX=100000000 timeout 120 python tools/r06_refscene_trace.py wcsph 40 200 2>&1 | grep "ms/step")
  b=$(NX=$nx SPHX_GROUP_BUILD_MAX=-1 timeout 120 python tools/r06_refscene_trace.py wcsph 40 200 2>&1 | grep "ms/step")
  echo "nx $nx: group: $a | lane: $b"
done
