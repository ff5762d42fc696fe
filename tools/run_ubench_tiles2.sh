#!/bin/bash
# Runs on the GPU box (via gpurun): particle-order experiment (modes qo) at 1 M and 10 M, and a counter pass (VALU issue, TA busy)
# over the quad walk and the all-pairs tiles at 10 M.
set -u
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 ./tools/ubench_tiles 88 20 qo > gpurun_out/ubench_order_1m.txt 2>&1
timeout 900 ./tools/ubench_tiles 190 10 qo > gpurun_out/ubench_order_10m.txt 2>&1
cd /tmp; rm -rf /tmp/ap_pmc
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE TA_TA_BUSY_sum --kernel-trace --output-format csv -d /tmp/ap_pmc -- $R/tools/ubench_tiles 190 1 qab > $R/gpurun_out/ubench_ap_pmc.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/ubench_ap_pmc.txt 2>&1
import csv, glob, collections
acc = collections.OrderedDict(); dur = {}
for f in glob.glob("/tmp/ap_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob("/tmp/ap_pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur.setdefault(r["Kernel_Name"][:60], [0.0, 0]); d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
print("counters per dispatch, 10,288,500 particles (VALU issue = SQ_INSTS_VALU x 2.3 clk / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))")
for k, d in acc.items():
    v = {c: t / n for c, (t, n) in d.items()}
    us = dur.get(k, [0, 1]); us = us[0] / max(us[1], 1)
    g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("%-62s %9.1f us  VALU instr %12.0f  issue %5.1f %%  TA busy %5.1f %%" % (
        k, us, v.get("SQ_INSTS_VALU", 0), 100 * v.get("SQ_INSTS_VALU", 0) * 2.3 / max(g * 1024, 1), 100 * v.get("TA_TA_BUSY_sum", 0) / max(g * 256, 1)))
PY
cat gpurun_out/ubench_order_1m.txt gpurun_out/ubench_order_10m.txt gpurun_out/ubench_ap_pmc.txt
