#!/bin/bash
# Runs on the GPU box (via gpurun): tools/ubench_brickflow at 10.3 M (and 1 M) particles, then PMC passes of the FLOW / Q4 kernels.
# usage: tools/run_brickflow.sh [variants=qbf] [pmc=1] [sizes="190 88"]
set -u
R=$PWD; export TMPDIR=/tmp
V=${1:-qbf}; PMC=${2:-1}; SIZES=${3:-"190 88"}
OUT=$R/gpurun_out/brickflow; mkdir -p $OUT
for NX in $SIZES; do
  for B in ubench_brickflow ubench_brickflow_noslp; do
    [ -x tools/$B ] || continue
    timeout 900 ./tools/$B $NX 10 $V > $OUT/${B}_$NX.txt 2>&1
  done
done
if [ "$PMC" = "1" ]; then
  cd /tmp
  P1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE TA_TA_BUSY_sum"
  P3="FETCH_SIZE TCP_TOTAL_CACHE_ACCESSES_sum"
  P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1)); rm -rf /tmp/bf_p$i
    timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/bf_p$i -- $R/tools/ubench_brickflow 190 2 ${V} > $OUT/pmc_p$i.log 2>&1
  done
  cd $R
  python3 - <<'PY' > $OUT/pmc_table.txt 2>&1
import csv, glob, collections, re
acc = collections.OrderedDict(); dur = collections.defaultdict(lambda: [0.0, 0])
def short(k):
    m = re.match(r"void (k_\w+)<([^>]*)>", k)
    return (m.group(1) + "<" + m.group(2) + ">") if m else k[:60]
for i in (1, 2, 3, 4):
    for f in glob.glob("/tmp/bf_p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("/tmp/bf_p%d/**/*kernel_trace.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            d = dur[short(r["Kernel_Name"])]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
for k, d in acc.items():
    print("%s   avg us under pmc %.1f" % (k, dur[k][0] / max(dur[k][1], 1)))
    for c, (t, n) in d.items():
        print("   %-30s %18.0f" % (c, t / n))
PY
fi
cat $OUT/ubench_brickflow_*.txt
[ "$PMC" = "1" ] && cat $OUT/pmc_table.txt
exit 0
