set -u
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or adaptive" 2>&1 | tail -6) > gpurun_out/r06_tail_tests.txt
cat gpurun_out/r06_tail_tests.txt
{ echo "## XCD-hierarchical barrier (default)"; SPHX_DFSPH_WINDOW=0 timeout 600 python tools/tail_probe.py 24 300 0 2>&1 | grep -v "^PBD\|^DFSPH\|^WCSPH\|per-kernel";
  echo "## r04 flat barrier (dfsph_tail_flat = 1)"; SPHX_DFSPH_WINDOW=0 SPHX_DFSPH_TAIL_FLAT=1 timeout 600 python tools/tail_probe.py 24 300 0 2>&1 | grep -v "^PBD\|^DFSPH\|^WCSPH\|per-kernel";
  echo "## default windows, XCD barrier"; timeout 600 python tools/tail_probe.py 24 300 0 2>&1 | grep -v "^PBD\|^DFSPH\|^WCSPH\|per-kernel"; } > gpurun_out/r06_tail_probe.txt 2>&1
cat gpurun_out/r06_tail_probe.txt
