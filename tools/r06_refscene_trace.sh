set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "pbd 250 50" "dfsph 250 50" "wcsph 250 100"; do
  set -- $cfg
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python $R/tools/r06_refscene_trace.py $1 $2 $3 > /tmp/tr.log 2>&1; grep "ms/step" /tmp/tr.log || tail -5 /tmp/tr.log
  f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
  python - "$f" $3 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
for r in rows[:12]:
    print("   %-70s calls %6s  avg %8.1f us  total %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
done
