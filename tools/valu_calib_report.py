"""Divides the SQ counters of a `rocprofv3 --pmc ... --kernel-trace -- tools/valu_calib` run by the instruction counts the
micro-benchmark is known to execute (tools/valu_calib.hip): prints, per kernel and waves/SIMD, SQ_INSTS_VALU per issued
wave-instruction, SQ_ACTIVE_INST_VALU per issued wave-instruction, and GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES per kernel
microsecond.   python tools/valu_calib_report.py <rocprof output dir> [trips=4096]"""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]; trips = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rows = []
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
trace = {}
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
per = defaultdict(dict)
for r in rows:
    per[(r["Dispatch_Id"], r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
print("%-10s %6s %14s %12s %14s %16s %14s %14s" % ("kernel", "w/SIMD", "wave-instr", "us", "INSTS_VALU/wi", "ACTIVE_VALU/wi", "GRBM/us", "SQ_BUSY/us"))
seen = defaultdict(int)
for (d, k, grid), c in sorted(per.items(), key=lambda kv: int(kv[0][0])):
    name = "dep" if "k_dep" in k else "indep" if "k_indep" in k else "packed" if "k_packed" in k else None
    if not name:
        continue
    waves = grid // 64
    wi = waves * trips * 64.0
    seen[(name, waves)] += 1
    if seen[(name, waves)] != 2:          # the timed launch (second of each pair)
        continue
    us = trace.get(d, float("nan"))
    print("%-10s %6d %14.0f %12.1f %14.4f %16.4f %14.1f %14.1f" % (name, waves // 1024, wi, us, c.get("SQ_INSTS_VALU", float("nan")) / wi,
          c.get("SQ_ACTIVE_INST_VALU", float("nan")) / wi, c.get("GRBM_GUI_ACTIVE", float("nan")) / us, c.get("SQ_BUSY_CYCLES", float("nan")) / us))
