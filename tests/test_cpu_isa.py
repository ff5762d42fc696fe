"""Static checks on the gfx950 code hipcc generates for the sweep kernels (cross-compiled, no GPU):
no scratch spills, full occupancy for the global-gather variants, and every neighbour gather of the
dominant kernel in the uniform-base + 32-bit-offset form ("Unified neighbour space")."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sweep_kernels_have_no_spills_and_uniform_base_gathers():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_report.py"), "dfsph"], check=True,
                         capture_output=True, text=True).stdout.splitlines()
    rows = [l for l in out if re.search(r"\s\d+\s+-?\d+\s+\d+\s+\d+\s+\d+\s+\d+$", l)]
    assert len(rows) > 20
    seen_rate = False
    for i, l in enumerate(out):
        m = re.search(r"^(.*?)\s+(\d+)\s+(-?\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)$", l)
        if not m:
            continue
        name, vgpr, occ, scratch, lds = m.group(1), int(m.group(2)), int(m.group(4)), int(m.group(5)), int(m.group(6))
        assert scratch == 0, "scratch spill in " + name
        if "k_rate<true, 2, false>" in name:          # the dominant kernel, global-gather variant
            seen_rate = True
            assert lds == 0 and occ >= 7 and vgpr <= 72, l
            mix = out[i + 1]
            assert "main loop" in mix
            assert re.search(r"gathers uniform-base 8, per-lane-base 0", mix), mix
            assert re.search(r"row loads 4", mix), mix
    assert seen_rate
