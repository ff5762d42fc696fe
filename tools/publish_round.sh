#!/bin/bash
# Copies what tools/collect_round.sh <tag> left in gpurun_out/ into profiles/ under the names the docs cite.
#   tools/publish_round.sh r04
set -eu
T=${1:-r04}
for V in "" strict; do
  S=gpurun_out/profile_summary_$T$V.txt
  [ -s $S ] || continue
  N=$([ -z "$V" ] && echo headline || echo strict)
  cp $S profiles/${T}_rocprofv3_dfsph10m_${N}_summary.txt
  # gpurun_out/ accumulates over calls: take the csv the summary itself names
  CSV=$(grep -o "[0-9]*_kernel_stats.csv" $S | head -1)
  cp "$(find gpurun_out/prof_$T$V -name "$CSV" | head -1)" profiles/${T}_rocprofv3_dfsph10m_${N}_kernel_stats.csv
done
cp gpurun_out/bench_${T}_loopback_1slabs.json profiles/${T}_bench_dfsph10m_loopback_1slab.json
cp gpurun_out/bench_${T}_loopback_8slabs.json profiles/${T}_bench_dfsph10m_loopback_8slabs_one_gpu.json
[ -s gpurun_out/bench_${T}_rcclself_8slabs.json ] && cp gpurun_out/bench_${T}_rcclself_8slabs.json profiles/${T}_bench_dfsph10m_rccl_self_8slabs_one_gpu.json
cp gpurun_out/pcie_$T.txt                     profiles/${T}_pcie_inclusive.txt
cp gpurun_out/probe_$T.txt                    profiles/${T}_probe_per_kernel_hipevents.txt
cp gpurun_out/small_$T.txt                    profiles/${T}_reference_scene_step_n.txt
cp gpurun_out/settle_${T}_190_adaptive.txt    profiles/${T}_settle_10m_adaptive.txt
cp gpurun_out/settle_${T}_88_fixed14.txt      profiles/${T}_settle_1m_fixed_1_4.txt
cp gpurun_out/big_$T.txt                      profiles/${T}_big_scenes.txt
cp gpurun_out/pytest_gpu_tail_$T.txt          profiles/${T}_pytest_gpu_tail.txt
python tools/make_traffic_json.py $T dfsph_nx190_tol > /dev/null
[ -s gpurun_out/traffic_${T}strict.json ] && python tools/make_traffic_json.py ${T}strict dfsph_nx190 > /dev/null
cp gpurun_out/bench_${T}_1gpu.json profiles/${T}_bench_dfsph10m_1gpu.json      # same box and call as the rocprofv3 summaries above
echo "published $T"
