# one GPU call: row-by-row rebuilds of PBD skin rows -- parity, then the landed regime with and without them
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "pbd or 2 or schedules or landing or stress or disordered" > gpurun_out/r06_partial_parity.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_partial_parity.log | tail -3
echo "--- reference scene PBD(20) and config 4 (1,022,208 particles, PBD(4)): free fall | landed, row-by-row rebuilds on / off"
for v in "" "SPHX_PBD_NO_PARTIAL=1"; do
  env $v timeout 300 python - <<'PY' 2>&1 | grep -v "^PBD:"
import sys, os
sys.path.insert(0, "cpp-fluid-particles_amd"); sys.path.insert(0, "tests")
import sphx, tuning_env
tuning_env.install(sphx)
for name, nx, iters, settle in (("reference scene PBD(20)", 24, 20, 250), ("config 4 PBD(4)", 88, 4, 300)):
    for arith in (0, 1):
        P, f, b = sphx.scene(nx); P.solver = sphx.PBD; P.dt = 0.002; P.pbd_iters = iters; P.reserved[3] = arith
        s = sphx.System(P, f, b); s.step_n(10)
        ff = min(s.step_n(20) / 20 for _ in range(2))
        s.step_n(settle); r0 = s.rows_stale()
        ms = min(s.step_n(50) / 50 for _ in range(2)); r1 = s.rows_stale()
        print("%s, arith %d, partial %s: free fall %.3f ms/step | landed %.3f ms/step, %.2f whole rebuilds per step" % (name, arith, "off" if os.environ.get("SPHX_PBD_NO_PARTIAL") else "on", ff, ms, (r1 - r0) / 100.0), flush=True)
        s.close()
PY
done
