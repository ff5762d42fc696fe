"""profiles/traffic.json from the counter passes of tools/profile_gpu.sh (gpurun_out/traffic_<tag>.json).
   python tools/make_traffic_json.py r05persistent dfsph_nx190_persistent"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
src = json.load(open(os.path.join(ROOT, "gpurun_out", "traffic_%s.json" % tag)))["k_rate_density"]
entry = {
    "kernel": "the density-error sweep k_rate_quad<true, 2, *> (computeDensityError_CUDA, quad-per-particle walk) as the %s leg of bench.py launches it, "
              "one launch at 10,288,500 particles" % key.rsplit("_", 1)[-1],
    "hbm_bytes_per_launch": src["hbm_bytes_fetch_x2"],
    "fetch_size_raw_bytes": src["FETCH_SIZE"], "write_size_bytes": src["WRITE_SIZE"],
    "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: gfx950 reports half of the fetched bytes; calibrated for THIS path's "
                  "access widths -- 16-byte and 4-byte streams, the quad walk's row read, 16-byte gathers -- in profiles/r04_ubench_tiles.txt: all x2); "
                  "WRITE_SIZE exact for full records, a 4-byte update inside a 16-byte record is tallied as its whole 64-byte line; separate --pmc passes",
    "valu_issue_frac": src.get("valu_issue_frac"),
    "valu_instr_per_simd_per_clk_raw": src.get("valu_instr_per_simd_per_clk_raw"),
    "valu_note": "SQ_INSTS_VALU x 2.3 clk / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); 2.3 clk per wave64 fp32 instruction calibrated in "
                 "profiles/r03_valu_calibration.txt (never clipped; the raw instructions per SIMD per clock are given beside it)",
    "sq_insts_valu_per_launch": src.get("SQ_INSTS_VALU"),
    "ta_busy_frac": src.get("ta_busy_frac"),
    "l1_hit_rate": 1.0 - src["TCP_TCC_READ_REQ_sum"] / src["TCP_TOTAL_CACHE_ACCESSES_sum"] if "TCP_TCC_READ_REQ_sum" in src else None,
    "l1_line_accesses_per_clk_per_cu": src.get("l1_line_accesses_per_clk_per_cu"),
    "l2_hit_rate": src["TCC_HIT_sum"] / (src["TCC_HIT_sum"] + src["TCC_MISS_sum"]) if "TCC_HIT_sum" in src else None,
    "avg_launch_us_rocprof": src.get("avg_launch_us_rocprof"),
    "source_hash": src["source_hash"],
    "source": "profiles/%s_rocprofv3_dfsph10m_%s_summary.txt (tools/profile_gpu.sh %s --arith %s)"
              % (tag[:3], key.rsplit("_", 1)[-1], tag, key.rsplit("_", 1)[-1]),
}
path = os.path.join(ROOT, "profiles", "traffic.json")
try:
    table = json.load(open(path))
except Exception:
    table = {}
table = {k: v for k, v in table.items() if v.get("source_hash") == entry["source_hash"]}     # entries of other source trees are stale
table[key] = entry
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
