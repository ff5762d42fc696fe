# one GPU call: the one-launch scans of the grid pass against the parity and slab suites, then the timings they were written for
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r06_chain_parity.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_chain_parity.log | tail -3
timeout 1500 python -m pytest tests/test_gpu_slab.py -x -q > gpurun_out/r06_chain_slab.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06_chain_slab.log | tail -3
timeout 300 python tools/r06_small_probe.py 2>&1 | grep -v "^PBD" > gpurun_out/r06_chain_small.txt; cat gpurun_out/r06_chain_small.txt
timeout 300 python tools/small_probe.py 2>&1 | grep -v "^PBD" | tail -12
for s in 1 8; do timeout 600 python bench.py --force-slab --slabs $s --arith tolerance --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slabs', $s, d['ms_per_step'])"; done | tee gpurun_out/r06_chain_slabbench.txt
