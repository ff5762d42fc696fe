#!/bin/bash
# Runs on the GPU box: kernel timeline of ONE graph-replayed step of the reference scene (adaptive DFSPH), with and without the loop tail.
#   bash tools/timeline_probe.sh   -> gpurun_out/timeline_{tail,gated}.txt
set -u
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/tl_run.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["R"], "cpp-fluid-particles_amd"))
import sphx
P, f, b = sphx.scene(24); P.solver = sphx.DFSPH
s = sphx.System(P, f, b); s.step(); s.step_n(40); s.close()
PY
for mode in tail gated; do
  rm -rf /tmp/tl_$mode
  if [ $mode = gated ]; then export SPHX_DFSPH_NO_TAIL=1; else unset SPHX_DFSPH_NO_TAIL; fi
  (cd /tmp && R=$R rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -- python /tmp/tl_run.py > /tmp/tl_$mode.log 2>&1)
  python3 - $mode <<'PY' > gpurun_out/timeline_$mode.txt
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/tl_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last complete step: from the last 'k_grid' / first kernel of a step to the end
names = [r[2] for r in rows]
starts = [k for k, nm in enumerate(names) if "cell_count" in nm or "k_count" in nm]
if len(starts) < 3: starts = [0, len(rows) // 2, len(rows)]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]; prev = t0
print("one step: %d launches, %.1f us from first start to last end" % (b - a, (rows[b - 1][1] - t0) / 1e3))
for k in range(a, b):
    s, e, nm = rows[k]
    print("%8.1f us  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm[:90]))
    prev = e
PY
done
head -3 gpurun_out/timeline_tail.txt; grep -c . gpurun_out/timeline_tail.txt gpurun_out/timeline_gated.txt
