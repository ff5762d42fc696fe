#!/bin/bash
# Runs on the GPU box: kernel timeline of the 10.3 M DFSPH(1,4) step in persistent-rows mode; one step WITH a row build and one without.
set -u
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/tp_run.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["R"], "cpp-fluid-particles_amd"))
import sphx
P, f, b = sphx.scene(int(os.environ.get("NX", "190"))); P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4; P.reserved[3] = 2
s = sphx.System(P, f, b); s.step(); s.step_n(30); print(s.persistent_stats()); s.close()
PY
rm -rf /tmp/tp
(cd /tmp && R=$R rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python /tmp/tp_run.py > /tmp/tp.log 2>&1)
python3 - <<'PY' > gpurun_out/timeline_persist.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/tp/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [k for k, r in enumerate(rows) if "k_persist_export" in r[2]]
for label, (a, b) in (("step A", (ends[-3] + 1, ends[-2] + 1)), ("step B", (ends[-2] + 1, ends[-1] + 1))):
    t0 = rows[a][0]; prev = t0; busy = 0
    print("%s: %d launches, %.1f us wall" % (label, b - a, (rows[b - 1][1] - t0) / 1e3))
    for k in range(a, b):
        s, e, nm = rows[k]; busy += e - s
        print("%9.1f us  dur %8.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm[:100]))
        prev = e
    print("   busy %.1f us" % (busy / 1e3))
PY
tail -3 /tmp/tp.log; wc -l gpurun_out/timeline_persist.txt
