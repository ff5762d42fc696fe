"""Condenses rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a text
summary and gpurun_out/traffic_<tag>.json (per-launch HBM bytes of the dominant kernel)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

VALU_CLK_PER_INSTR = 2.3      # profiles/r03_valu_calibration.txt
out, tag = sys.argv[1], sys.argv[2]
bench_args = sys.argv[3:]


def is_density_rate(name, warm2=False):
    """the dominant kernel: computeDensityError's sweep, k_rate<true, W, ...> or its quad-per-particle variant
    k_rate_quad<true, W> (W = 2: the iterations after the first)"""
    import re
    m = re.search(r"k_rate(_quad)?<(true|\(bool\)1), (\d)", name)
    return bool(m) and (not warm2 or m.group(3) == "2")


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("sphx::", "").replace("(anonymous namespace)::", "")
    return name[:110]


print("rocprofv3 summary tag=%s bench_args=%s" % (tag, " ".join(bench_args)))
ks = find("stats", "*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    print("\n== kernel stats (rocprofv3 --kernel-trace --stats), %s" % os.path.basename(ks))
    print("%-112s %8s %12s %8s" % ("kernel", "calls", "avg_us", "pct"))
    for r in rows[:40]:
        print("%-112s %8s %12.2f %8s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
else:
    print("no kernel_stats.csv found")

traffic = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        print("no counter csv for", ctr)
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    print("\n== %s per dispatch (raw counter units: KiB; gfx950 FETCH_SIZE under-reads wide streams by 2x)" % ctr)
    for k, (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:25]:
        print("%-112s %6d launches  avg %12.1f KiB" % (k, n, tot / n))
        if is_density_rate(k):
            traffic.setdefault("k_rate_density", {})[ctr] = tot / n * 1024.0
# VALU / cache passes: per-dispatch averages of the sweep kernels
extra = {}
for sub in ("valu", "cache"):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(cc)):
        k = short(r["Kernel_Name"])
        if "k_rate" in k or "k_run_op" in k or "k_build_list" in k or "k_dfsph_head" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print("\n== %s pass, per-dispatch averages" % sub)
    for k, d in acc.items():
        print(k)
        for cname, (tot, cnt) in sorted(d.items()):
            print("   %-36s %18.1f  (avg of %d)" % (cname, tot / cnt, cnt))
            if is_density_rate(k, warm2=True):
                extra[cname] = tot / cnt
if traffic:
    kr = traffic["k_rate_density"]
    kr.update(extra)
    if "SQ_INSTS_VALU" in extra and extra.get("GRBM_GUI_ACTIVE"):
        # Calibration (profiles/r03_valu_calibration.txt, tools/valu_calib.hip): SQ_INSTS_VALU = 1 per issued wave64 VALU
        # instruction; GRBM_GUI_ACTIVE / 8 XCDs = elapsed shader clocks; a non-packed fp32 instruction occupies its SIMD for
        # 2.3 clocks when nothing else limits it (packed ones twice that; not distinguished by the counter);
        # SQ_ACTIVE_INST_VALU reads like SQ_INSTS_VALU on gfx950 and is NOT a busy-cycle count (r01/r02 used it as one).
        cycles = extra["GRBM_GUI_ACTIVE"] / 8.0
        kr["kernel_cycles"] = cycles
        kr["valu_clocks_per_instruction_calibrated"] = VALU_CLK_PER_INSTR
        kr["valu_instr_per_simd_per_clk_raw"] = extra["SQ_INSTS_VALU"] / (cycles * 1024.0)
        kr["valu_issue_frac"] = extra["SQ_INSTS_VALU"] * VALU_CLK_PER_INSTR / (cycles * 1024.0)
        stats = find("stats", "*kernel_stats.csv")
        us = None
        if stats:
            for r in csv.DictReader(open(stats)):
                if is_density_rate(r["Name"], warm2=True):
                    us = float(r["AverageNs"]) / 1e3
        if us:
            kr["avg_launch_us_rocprof"] = us
        if "TA_TA_BUSY_sum" in extra:
            kr["ta_busy_frac"] = extra["TA_TA_BUSY_sum"] / 256.0 / cycles          # 256 CUs, one texture-address unit each
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in extra:
            kr["l1_line_accesses_per_clk_per_cu"] = extra["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cycles
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from srchash import engine_source_hash
    kr["source_hash"] = engine_source_hash()
    fetch = kr.get("FETCH_SIZE"); write = kr.get("WRITE_SIZE")
    if fetch is not None and write is not None:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of wide coalesced streaming reads on gfx950; this
        # kernel's reads are mostly 16-byte gathers + 4-byte coalesced rows, so both raw and x2 are reported
        kr["hbm_bytes_raw"] = fetch + write
        kr["hbm_bytes_fetch_x2"] = 2 * fetch + write
    json.dump(traffic, open(os.path.join(os.path.dirname(out), "traffic_%s.json" % tag), "w"), indent=1)
    print("\ntraffic:", json.dumps(traffic))
