#!/bin/bash
# counters of the lane-per-particle row builder at 263,424 (config 2) and 10,288,500 particles: why is it half as efficient at the smaller size?
R=$PWD; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/builder_pmc; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/target.py <<PY
import sys, os
sys.path.insert(0, os.path.join("$R", "cpp-fluid-particles_amd"))
import sphx
nx = int(os.environ["NX"])
P, f, b = sphx.scene(nx); P.solver = sphx.WCSPH; P.dt = 0.001
sphx.set_tuning(no_graph=1)
s = sphx.System(P, f, b)
for _ in range(6): s.step()
s.close()
PY
for nx in 56 190; do
NX=$nx timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/a$nx -- python /tmp/target.py > $OUT/a$nx.log 2>&1
NX=$nx timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE TA_TA_BUSY_sum --kernel-trace --output-format csv -d $OUT/b$nx -- python /tmp/target.py > $OUT/b$nx.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for nx in (56, 190):
    acc = collections.defaultdict(lambda: [0.0, 0]); dur = [0.0, 0]
    for f in glob.glob("gpurun_out/builder_pmc/[ab]%d/**/*counter_collection.csv" % nx, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_build_list<false>" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob("gpurun_out/builder_pmc/a%d/**/*kernel_trace.csv" % nx, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_build_list<false>" in r["Kernel_Name"]:
                dur[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[1] += 1
    print("nx %d: k_build_list<false> avg %.1f us under counters (%d launches)" % (nx, dur[0] / max(dur[1], 1), dur[1]))
    for c, (t, n) in sorted(acc.items()):
        print("   %-28s %16.1f" % (c, t / n))
PY
