"""Per-kernel counter table of a tools/profile_gpu.sh run (gpurun_out/prof_<tag>/): average launch time (kernel stats pass), calibrated
VALU issue, TA busy, vector-L1 accesses per clock per CU, L1 / L2 hit rates and HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE).
   python tools/pmc_table.py r04 [r04strict ...]"""
import csv, glob, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("k_rate", "k_run_op", "k_build_list", "k_dfsph_head")
def short(n):
    n = n.replace("sphx::", "")
    n = re.sub(r"\(.*$", "", n)
    return n.replace("void ", "")[:74]
for tag in sys.argv[1:] or ["r04"]:
    out = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    per = defaultdict(dict)
    for sub in ("fetch", "write", "valu", "cache"):
        # (gpurun_out/ accumulates over calls: only the newest file of a pass belongs to the last collection)
        for f in sorted(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if any(h in k for h in HOT):
                    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
            for k, d in acc.items():
                for c, (t, n) in d.items():
                    per[k][c] = t / n
    us = {}
    for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            us[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    print("== %s: counters per launch at 10,288,500 particles (VALU issue = SQ_INSTS_VALU x 2.3 clk / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); HBM = FETCH_SIZE x 2 + WRITE_SIZE)" % tag)
    print("%-76s %8s %6s %6s %8s %6s %6s %9s" % ("kernel", "us", "VALU%", "TA%", "L1/clk/CU", "L1hit%", "L2hit%", "HBM MB"))
    for k in sorted(per, key=lambda k: -us.get(k, (0, 0))[0] * us.get(k, (0, 0))[1]):
        d = per[k]
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        f = lambda v: ("%6.1f" % v) if v is not None else "   n/a"
        valu = 100 * d["SQ_INSTS_VALU"] * 2.3 / (cyc * 1024) if cyc and "SQ_INSTS_VALU" in d else None
        ta = 100 * d["TA_TA_BUSY_sum"] / 256.0 / cyc if cyc and "TA_TA_BUSY_sum" in d else None
        l1 = d["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cyc if cyc and "TCP_TOTAL_CACHE_ACCESSES_sum" in d else None
        l1h = 100 * (1 - d["TCP_TCC_READ_REQ_sum"] / d["TCP_TOTAL_CACHE_ACCESSES_sum"]) if d.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None
        l2h = 100 * d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]) if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) else None
        hbm = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e6 if "FETCH_SIZE" in d and "WRITE_SIZE" in d else None
        print("%-76s %8.1f %s %s %8s %s %s %9s" % (k, us.get(k, (0, 0))[0], f(valu), f(ta), ("%8.2f" % l1) if l1 is not None else "     n/a", f(l1h), f(l2h),
                                              ("%9.0f" % hbm) if hbm is not None else "      n/a"))
    print()
