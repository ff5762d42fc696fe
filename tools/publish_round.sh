#!/bin/bash
# Copies what tools/collect_round.sh <tag> left in gpurun_out/ into profiles/ under the names the docs cite.
#   tools/publish_round.sh r05
set -eu
T=${1:-r05}
for A in persistent tolerance strict; do
  S=gpurun_out/profile_summary_$T$A.txt
  [ -s $S ] || continue
  N=$([ $A = persistent ] && echo headline || echo $A)
  cp $S profiles/${T}_rocprofv3_dfsph10m_${N}_summary.txt
  # gpurun_out/ accumulates over calls: take the csv the summary itself names
  CSV=$(grep -o "[0-9]*_kernel_stats.csv" $S | head -1)
  cp "$(find gpurun_out/prof_$T$A -name "$CSV" | head -1)" profiles/${T}_rocprofv3_dfsph10m_${N}_kernel_stats.csv
  [ -s gpurun_out/traffic_$T$A.json ] && python tools/make_traffic_json.py $T$A dfsph_nx190_$A > /dev/null
done
for A in tolerance strict; do
  cp gpurun_out/bench_${T}_loopback_1slabs_$A.json profiles/${T}_bench_dfsph10m_loopback_1slab_$A.json
  cp gpurun_out/bench_${T}_loopback_8slabs_$A.json profiles/${T}_bench_dfsph10m_loopback_8slabs_one_gpu_$A.json
done
[ -s gpurun_out/bench_${T}_rcclself_8slabs.json ] && cp gpurun_out/bench_${T}_rcclself_8slabs.json profiles/${T}_bench_dfsph10m_rccl_self_8slabs_one_gpu.json
[ -s gpurun_out/bench_${T}_rcclself_8slabs_edgehigh.json ] && cp gpurun_out/bench_${T}_rcclself_8slabs_edgehigh.json profiles/${T}_bench_dfsph10m_rccl_self_8slabs_one_gpu_edge_high.json
for A in tolerance strict; do [ -s gpurun_out/bench_${T}_plain_$A.json ] && cp gpurun_out/bench_${T}_plain_$A.json profiles/${T}_bench_dfsph10m_plain_${A}_same_call.json; done
cp gpurun_out/pcie_$T.txt                     profiles/${T}_pcie_inclusive.txt
cp gpurun_out/probe_$T.txt                    profiles/${T}_probe_per_kernel_hipevents.txt
cp gpurun_out/small_$T.txt                    profiles/${T}_reference_scene_step_n.txt
[ -s gpurun_out/small_configs_$T.txt ] && cp gpurun_out/small_configs_$T.txt profiles/${T}_small_configs_per_kernel.txt
cp gpurun_out/slab_probe_$T.txt               profiles/${T}_slab_probe_step.txt
cp gpurun_out/big_$T.txt                      profiles/${T}_big_scenes.txt
for k in parity tolerance persistent slab slab_tolerance; do [ -s gpurun_out/stress_${T}_$k.txt ] && cp gpurun_out/stress_${T}_$k.txt profiles/${T}_stress_$k.txt; done
cp gpurun_out/pytest_gpu_tail_$T.txt          profiles/${T}_pytest_gpu_tail.txt
cp gpurun_out/bench_${T}_1gpu.json profiles/${T}_bench_dfsph10m_1gpu.json      # same box and call as the rocprofv3 summaries above
echo "published $T"
