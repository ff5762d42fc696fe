#!/bin/bash
# PMC passes for the 1M DFSPH probe; condensed per-kernel averages for the sweep kernels
R=$PWD; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python $R/tools/probe_step.py dfsph1m > $OUT/$1.log 2>&1; }
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run p2 "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32"
run p3 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run p4 "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
run p5 "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
cd $R
python - <<'PY'
import csv, glob, collections, os
out = "gpurun_out/pmc_probe"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "OpCorrect<true>" in k or "k_rate<true, 2" in k or "k_build_list" in k or "OpSurface" in k:
            a = acc[k[:70]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/pmc_probe_summary.txt", "w") as fo:
    for k, d in acc.items():
        fo.write(k + "\n")
        for c, (t, n) in sorted(d.items()):
            fo.write("   %-40s %16.1f  (avg over %d dispatches)\n" % (c, t / n, n))
print(open("gpurun_out/pmc_probe_summary.txt").read())
PY
find $OUT -name "*.csv" -size +2M -delete
