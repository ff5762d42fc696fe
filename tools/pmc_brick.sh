#!/bin/bash
# counters of the brick kernels vs the quad kernels (tolerance arithmetic): tools/pmc_brick.sh
R=$PWD; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_brick; rm -rf $OUT; mkdir -p $OUT
for mode in brick quad; do
  if [ $mode = quad ]; then export SPHX_BRICK=0; else unset SPHX_BRICK; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/$mode -- python $R/tools/pmc_brick_target.py > $OUT/$mode.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for mode in ("brick", "quad"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])); dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/pmc_brick/%s/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_rate" in k and "true, 2" in k or "OpCorrect<true>" in k or "k_build" in k:
                a = acc[k[:70]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("gpurun_out/pmc_brick/%s/**/*kernel_trace.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if k[:70] in acc:
                d = dur[k[:70]]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
    for k, d in acc.items():
        print(mode, k, "avg us (under pmc) %.1f" % (dur[k][0] / max(dur[k][1], 1)))
        for c, (t, n) in sorted(d.items()):
            print("   %-26s %16.0f" % (c, t / n))
PY
