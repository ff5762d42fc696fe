#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/rNN_* is made from, in one call.
#   tools/collect_round.sh r05
set -u
TAG=${1:-r05}
R=$PWD
mkdir -p gpurun_out
# the GPU suite first: a red suite ends the call here (the rest of the call would measure a broken tree)
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "PBD:\|amdgpu\|Could not read\|iommu" | tail -4 > gpurun_out/pytest_gpu_tail_$TAG.txt
cat gpurun_out/pytest_gpu_tail_$TAG.txt
grep -q " passed" gpurun_out/pytest_gpu_tail_$TAG.txt && ! grep -q "failed\|error" gpurun_out/pytest_gpu_tail_$TAG.txt || { echo "GPU suite not green: stopping"; exit 1; }
# rocprofv3 kernel stats + PMC passes of the three legs of bench.py (persistent = the headline, tolerance, strict); the counters just
# collected become profiles/traffic.json of THIS copy, so that the bench line below carries them (hash-checked) and comes from the same
# box as the rocprofv3 summaries; publish_round.sh regenerates the same file in the repository
for A in persistent tolerance strict; do
  bash tools/profile_gpu.sh $TAG$A --arith $A > gpurun_out/profile_$TAG$A.log 2>&1
  python tools/make_traffic_json.py $TAG$A dfsph_nx190_$A > /dev/null 2>&1
done
python bench.py > gpurun_out/bench_${TAG}_1gpu.json 2> gpurun_out/bench_${TAG}_1gpu.err
python tools/probe_step.py wcsph263k dfsph1m pbd1m dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" > gpurun_out/probe_$TAG.txt
TOL=1 python tools/probe_step.py dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" | sed 's/^dfsph10m/dfsph10m(tolerance)/' >> gpurun_out/probe_$TAG.txt
TOL=2 python tools/probe_step.py dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" | sed 's/^dfsph10m/dfsph10m(persistent)/' >> gpurun_out/probe_$TAG.txt
(echo "# reference scene (20,736 particles), default solver settings, step_n batches of 100 behind 10 steps: free fall | landing | landed.  strict arithmetic:"; python tools/small_probe.py 2>/dev/null | grep -v "amdgpu\|^PBD"; echo "# headline arithmetic (TOL=2):"; TOL=2 python tools/small_probe.py 2>/dev/null | grep -v "amdgpu\|^PBD") > gpurun_out/small_$TAG.txt
python tools/r06_small_probe.py 2>/dev/null | grep -v "amdgpu\|^PBD" > gpurun_out/small_configs_$TAG.txt
python tools/pcie_probe.py 2>/dev/null | grep -v amdgpu > gpurun_out/pcie_$TAG.txt
for A in tolerance strict; do for s in 1 8; do python bench.py --force-slab --slabs $s --arith $A --steps 20 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_${TAG}_loopback_${s}slabs_$A.json; done; done
python bench.py --force-slab --slabs 8 --arith tolerance --slab-transport rccl --steps 20 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_${TAG}_rcclself_8slabs.json
for A in tolerance strict; do python bench.py --arith $A --steps 20 --no-cpu-baseline --no-extra-legs 2>/dev/null > gpurun_out/bench_${TAG}_plain_$A.json; done
(python tools/slab_probe_step.py 190 1 1; python tools/slab_probe_step.py 190 8 1) 2>/dev/null | grep -v "amdgpu\|^PBD" > gpurun_out/slab_probe_$TAG.txt
python tools/big_probe.py 190,320,400 0 2>/dev/null | grep "^nx" > gpurun_out/big_${TAG}.txt
bash tools/stress_round.sh $TAG > gpurun_out/stress_$TAG.log 2>&1
cat gpurun_out/pytest_gpu_tail_$TAG.txt; cat gpurun_out/pcie_$TAG.txt; grep "ms/step" gpurun_out/probe_$TAG.txt; cat gpurun_out/small_$TAG.txt
