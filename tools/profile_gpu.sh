#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and the two PMC passes for bench.py,
# then condenses them into gpurun_out/profile_summary_<tag>.txt / traffic_<tag>.json.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=$PWD
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/valu -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > $OUT/valu.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum --kernel-trace --output-format csv -d $OUT/cache -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs "$@" > $OUT/cache.log 2>&1
cd $R
python tools/profile_summarize.py $OUT $TAG "$@" > gpurun_out/profile_summary_$TAG.txt 2>&1
# keep the merge small: raw traces stay on the box except the stats csv
find $OUT -name "*.csv" -size +8M -delete
tail -40 gpurun_out/profile_summary_$TAG.txt
