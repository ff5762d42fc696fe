#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_lite; rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python $R/tools/pmc_lite_target.py > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE TA_TA_BUSY_sum --kernel-trace --output-format csv -d $OUT/p2 -- python $R/tools/pmc_lite_target.py > $OUT/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_lite/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "OpCorrect<true>" in k or "k_rate<true, 2" in k:
            a = acc[k[:60]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob("gpurun_out/pmc_lite/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "OpCorrect<true>" in k or "k_rate<true, 2" in k:
            d = dur[k[:60]]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
for k, d in acc.items():
    print(k, "avg us (under pmc)", dur[k][0] / max(dur[k][1], 1))
    for c, (t, n) in sorted(d.items()):
        print("   %-30s %14.1f (%d)" % (c, t / n, n))
PY
