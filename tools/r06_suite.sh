set -u
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1) 2> gpurun_out/r06_gpu_suite_time.txt
(grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r06_gpu_suite.log | tail -6; cat gpurun_out/r06_gpu_suite_time.txt) > gpurun_out/r06_gpu_suite.txt
cat gpurun_out/r06_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(time python bench.py --steps 20 --warmup 5 2> gpurun_out/r06_bench.log > gpurun_out/r06_bench.json) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus", "dtype")}); print("roofline", d.get("roofline")); cb = d.get("cpu_baseline", {}); print("cpu_baseline", {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "extrapolated")})
PY
