"""Per-slab-step kernel budget of a loopback slab run from a rocprofv3 --kernel-trace --stats csv (GPU box, see tools/README.md):
python tools/slab_profile_report.py <kernel_stats.csv> <slabs> <steps incl. constructor and warm-up>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
slabs, steps = int(sys.argv[2]), int(sys.argv[3])
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time per step %.3f ms, per slab-step %.3f ms; launches per slab-step %.1f" % (tot / steps / 1e6, tot / steps / slabs / 1e6, calls / steps / slabs))
print("%-78s %9s %9s %9s" % ("kernel", "calls/ss", "avg us", "us/ss"))
for r in rows[:40]:
    c = int(r["Calls"]) / steps / slabs
    print("%-78s %9.1f %9.1f %9.1f" % (r["Name"][:78], c, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / steps / slabs / 1e3))
