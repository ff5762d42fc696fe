#!/bin/bash
# Runs on the GPU box (via gpurun): tools/ubench_tiles at 1 M and 10 M particles, then the calibration dispatches under
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes).  Output: gpurun_out/ubench_tiles_*.txt
set -u
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
MODES=${1:-qabB}
timeout 300 ./tools/ubench_tiles 88 20 $MODES > gpurun_out/ubench_tiles_1m.txt 2>&1
timeout 300 ./tools/ubench_tiles 190 10 $MODES > gpurun_out/ubench_tiles_10m.txt 2>&1
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cal_$C -- $R/tools/ubench_tiles 190 1 qc > $R/gpurun_out/ubench_tiles_cal_$C.log 2>&1
done
cd $R
python3 - <<'PY' > gpurun_out/ubench_tiles_cal.txt 2>&1
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.OrderedDict()
    for f in glob.glob("/tmp/cal_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"][:70]
            a = acc.setdefault(k, [0.0, 0, []]); a[0] += float(r["Counter_Value"]); a[1] += 1; a[2].append(float(r["Counter_Value"]))
    print("== %s per dispatch (raw counter value; rocprofv3 reports KiB)" % ctr)
    for k, (t, n, vals) in acc.items():
        print("%-72s %3d dispatches  avg %14.1f  last %14.1f" % (k, n, t / n, vals[-1]))
PY
grep -h "CAL\|reads\|writes\|touches" gpurun_out/ubench_tiles_cal_FETCH_SIZE.log >> gpurun_out/ubench_tiles_cal.txt
cat gpurun_out/ubench_tiles_1m.txt gpurun_out/ubench_tiles_10m.txt gpurun_out/ubench_tiles_cal.txt
