#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/rNN_* is made from, in one call.
#   tools/collect_round.sh r04
set -u
TAG=${1:-r04}
R=$PWD
mkdir -p gpurun_out
# rocprofv3 kernel stats + PMC passes of the HEADLINE leg (bench.py's default: tolerance arithmetic, persistent rows) and of the strict leg;
# the counters just collected become profiles/traffic.json of THIS copy, so that the bench line below carries them (hash-checked)
# and comes from the same box as the rocprofv3 summaries; publish_round.sh regenerates the same file in the repository
bash tools/profile_gpu.sh $TAG --arith persistent > gpurun_out/profile_$TAG.log 2>&1
python tools/make_traffic_json.py $TAG dfsph_nx190_tol > /dev/null 2>&1
bash tools/profile_gpu.sh ${TAG}strict --arith strict > gpurun_out/profile_${TAG}strict.log 2>&1
python tools/make_traffic_json.py ${TAG}strict dfsph_nx190 > /dev/null 2>&1
python bench.py > gpurun_out/bench_${TAG}_1gpu.json 2> gpurun_out/bench_${TAG}_1gpu.err
python tools/probe_step.py wcsph263k dfsph1m pbd1m dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" > gpurun_out/probe_$TAG.txt
TOL=1 python tools/probe_step.py dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" | sed 's/^dfsph10m/dfsph10m(tolerance)/' >> gpurun_out/probe_$TAG.txt
TOL=2 python tools/probe_step.py dfsph10m 2>/dev/null | grep -v "amdgpu\|^PBD" | sed 's/^dfsph10m/dfsph10m(persistent)/' >> gpurun_out/probe_$TAG.txt
python tools/small_probe.py 2>/dev/null | grep -v "amdgpu\|^PBD" > gpurun_out/small_$TAG.txt
python tools/pcie_probe.py 2>/dev/null | grep -v amdgpu > gpurun_out/pcie_$TAG.txt
for s in 1 8; do python bench.py --force-slab --slabs $s --steps 20 2>/dev/null > gpurun_out/bench_${TAG}_loopback_${s}slabs.json; done
python bench.py --force-slab --slabs 8 --slab-transport rccl --steps 20 2>/dev/null > gpurun_out/bench_${TAG}_rcclself_8slabs.json
python tools/settle_probe.py 190 350 -1 -1 2>/dev/null | grep -v amdgpu > gpurun_out/settle_${TAG}_190_adaptive.txt
python tools/settle_probe.py 88 450 1 4 2>/dev/null | grep -v amdgpu > gpurun_out/settle_${TAG}_88_fixed14.txt
python tools/big_probe.py 190,320,400 0 2>/dev/null | grep "^nx" > gpurun_out/big_${TAG}.txt
python -m pytest tests -m gpu -q 2>&1 | grep -v "PBD:\|amdgpu\|Could not read\|iommu" | tail -4 > gpurun_out/pytest_gpu_tail_$TAG.txt
cat gpurun_out/pytest_gpu_tail_$TAG.txt; cat gpurun_out/pcie_$TAG.txt; grep "ms/step" gpurun_out/probe_$TAG.txt; cat gpurun_out/small_$TAG.txt
