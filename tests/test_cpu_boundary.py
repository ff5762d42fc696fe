"""CPU tests (no GPU) of the drop-in boundary: the C ABI library loads and exports every symbol the
header declares, the host-side pieces (scene generator, parameter block) agree with the oracle,
and the product refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal


def _declared(header="sphx_c.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sphx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sphx):
    L = sphx.lib()
    names = _declared()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(L, nm), "libsphx.so does not export %s declared in include/sphx_c.h" % nm
    assert sorted(sphx.EXPORTS) == names, "sphx.py EXPORTS out of date with include/sphx_c.h"
    slab = [nm for nm in _declared("sphx_slab.h") if nm.startswith("sphx_slab_") and nm != "sphx_slab_group"]
    assert sorted(sphx.SLAB_EXPORTS) == slab, "sphx.py SLAB_EXPORTS out of date with include/sphx_slab.h"
    for nm in slab:
        assert hasattr(L, nm), "libsphx.so does not export %s declared in include/sphx_slab.h" % nm
    for nm in sphx.REFERENCE_EXPORTS:       # the reference's own extern "C" symbol (vbo.cu:46-51)
        assert hasattr(L, nm), "libsphx.so does not export the reference symbol %s" % nm


def test_param_block_layouts_match(sphx, oracle):
    assert C.sizeof(sphx.Params) == C.sizeof(oracle.Params) == sphx.lib().sphx_sizeof_params()
    assert [f[0] for f in sphx.Params._fields_] == [f[0] for f in oracle.Params._fields_]


@pytest.mark.parametrize("nx", [8, 24, 40])
def test_scene_generators_agree(sphx, oracle, nx):
    """the product's scene generator (C++) and the oracle's (C) are independent restatements of
    main.cpp:54-117; they must agree bit for bit."""
    Pg, fg, bg = sphx.scene(nx)
    Po, fo, bo = oracle.scene(nx)
    assert_bit_equal(fg, fo, "fluid"); assert_bit_equal(bg, bo, "boundary")
    for name, _ in Pg._fields_:
        a, b = getattr(Pg, name), getattr(Po, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name


def test_scene_rejects_bad_sizes(sphx):
    with pytest.raises(sphx.SphxError):
        sphx.scene(7)
    with pytest.raises(sphx.SphxError):
        sphx.scene(0)


def test_no_device_means_loud_failure(sphx):
    """there is no CPU path: on a box without a GPU, creating a system must raise."""
    if sphx.device_count() > 0:
        pytest.skip("a HIP device is present")
    P, fluid, boundary = sphx.scene(8)
    with pytest.raises(sphx.SphxError) as e:
        sphx.System(P, fluid, boundary)
    assert "no HIP device" in str(e.value)
    with pytest.raises(sphx.SphxError):
        sphx.eval_kernels(np.zeros((4, 3), np.float32), 0.04)


def test_create_validates_arguments(sphx):
    P, fluid, boundary = sphx.scene(8)
    P.pow7_mode = 1
    with pytest.raises(sphx.SphxError):
        sphx.System(P, fluid, boundary)
    P.pow7_mode = 0
    P.cells[0] = 0
    with pytest.raises(sphx.SphxError):
        sphx.System(P, fluid, boundary)


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under the package or include/ may mention it."""
    pkg = os.path.join(ROOT, "cpp-fluid-particles_amd")
    for base in (pkg, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            if "build" in dirpath:
                continue
            for f in files:
                if f.endswith((".hip", ".hpp", ".h", ".py", ".cpp")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "sph_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_fault_hooks_live_in_the_test_build_only():
    """the fault injection the slab tests use (SPHX_SLAB_FAULT) is compiled into tests/libsphx_hooks.so (-DSPHX_TEST_HOOKS), never
    into the product library"""
    product = os.path.join(ROOT, "cpp-fluid-particles_amd", "libsphx.so")
    hooks = os.path.join(ROOT, "tests", "libsphx_hooks.so")
    assert os.path.exists(product) and os.path.exists(hooks), "built by __graft_entry__.build()"
    for hook in (b"SPHX_SLAB_FAULT", b"SPHX_DFSPH_TAIL_OVERSUBSCRIBE"):
        assert hook not in open(product, "rb").read(), hook
        assert hook in open(hooks, "rb").read(), hook


def test_obstacle_samplers(sphx):
    """host-side boundary samplers for static obstacles (§8f-4): box lattice equals an independent numpy
    restatement, sphere and triangle samples lie on their surfaces and are nowhere sparser than the spacing"""
    lo, hi, h = np.float32([0.40, 0.0, 0.10]), np.float32([0.46, 0.15, 0.40]), np.float32(0.02)
    got = sphx.sample_box(lo, hi, h)
    n = [max(1, int(np.ceil((hi[a] - lo[a]) / h - 1e-4))) for a in range(3)]
    ax = [lo[a] + (hi[a] - lo[a]) * (np.arange(n[a] + 1, dtype=np.float32) / np.float32(n[a])) for a in range(3)]
    want = []
    for a in range(n[0] + 1):
        for b in range(n[1] + 1):
            want += [(ax[0][a], ax[1][b], lo[2]), (ax[0][a], ax[1][b], hi[2])]
    for a in range(n[0] + 1):
        for c in range(1, n[2]):
            want += [(ax[0][a], lo[1], ax[2][c]), (ax[0][a], hi[1], ax[2][c])]
    for b in range(1, n[1]):
        for c in range(1, n[2]):
            want += [(lo[0], ax[1][b], ax[2][c]), (hi[0], ax[1][b], ax[2][c])]
    assert_bit_equal(got, np.array(want, np.float32), "box sampler")
    assert len(np.unique(got, axis=0)) == len(got), "edges and corners exactly once"

    c, r = np.float32([0.25, 0.3, 0.25]), np.float32(0.05)
    s = sphx.sample_sphere(c, r, 0.01)
    assert np.abs(np.linalg.norm(s - c, axis=1) - r).max() < 1e-6
    d = np.linalg.norm(s[:, None, :] - s[None, :, :], axis=2) + np.eye(len(s)) * 9
    assert d.min(axis=1).max() <= 0.0101 and d.min() > 0.002, "no holes wider than the spacing, no duplicates"

    tri = np.float32([[0, 0, 0, 0.1, 0, 0, 0, 0.1, 0.02], [0.1, 0, 0, 0.1, 0.1, 0.02, 0, 0.1, 0.02]])
    t = sphx.sample_triangles(tri, 0.02)
    nrm = np.cross(tri[0, 3:6] - tri[0, 0:3], tri[0, 6:9] - tri[0, 0:3]); nrm /= np.linalg.norm(nrm)
    on0 = np.abs((t - tri[0, 0:3]) @ nrm) < 1e-6
    nrm1 = np.cross(tri[1, 3:6] - tri[1, 0:3], tri[1, 6:9] - tri[1, 0:3]); nrm1 /= np.linalg.norm(nrm1)
    on1 = np.abs((t - tri[1, 0:3]) @ nrm1) < 1e-6
    assert (on0 | on1).all()
    d = np.linalg.norm(t[:, None, :] - t[None, :, :], axis=2) + np.eye(len(t)) * 9
    assert d.min() > 0.02 / 8 and d.min(axis=1).max() <= 0.0205
    with pytest.raises(sphx.SphxError):
        sphx.sample_sphere(c, -1.0, 0.01)
    # far from the origin the dedupe must neither alias distinct points (no coordinate masking) nor miss the shared
    # edge because two copies straddle a lattice-cell border; a box that is flat along an axis has one face, not two coincident ones
    far = tri.copy(); far[:, 0::3] += np.float32(3000.0); far[:, 2::3] -= np.float32(2500.0)
    tf = sphx.sample_triangles(far, 0.02)
    assert len(tf) == len(t), "the same two triangles, translated by (3000, 0, -2500): %d vs %d points" % (len(tf), len(t))
    plate = sphx.sample_box(np.float32([0.1, 0.2, 0.1]), np.float32([0.3, 0.2, 0.3]), 0.02)
    assert len(np.unique(plate, axis=0)) == len(plate) and len(plate) == 11 * 11, "a flat box is ONE layer of points"
    tiny = sphx.lib().sphx_sample_box          # a spacing that would overflow the interval count is bounded, not UB: only counted
    import ctypes as C
    cnt = C.c_int()
    rc = tiny((C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1, 1, 1), C.c_float(1e-30), None, 0, C.byref(cnt))
    assert rc == -1, "a spacing that asks for more than 1e9 points is refused at once"


def test_bench_refuses_cpu_and_checks_traffic_provenance(tmp_path, monkeypatch):
    """bench.py has no CPU path (it must fail loudly without a HIP device), and it only reports the committed PMC
    traffic figure when that figure was measured on the source tree the library was built from"""
    import importlib.util
    import json
    import subprocess
    import sys
    import sphx as S
    if S.device_count() == 0:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
        assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import engine_source_hash
    h = engine_source_hash()
    assert len(h) == 16 and h == engine_source_hash()
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    # one entry per arithmetic contract of the headline workload (r05: keyed by the leg that launched the kernel, not by a guess of its instantiation)
    assert set(table) <= {"dfsph_nx190_strict", "dfsph_nx190_tolerance", "dfsph_nx190_persistent"} and "dfsph_nx190_strict" in table
    entry = table["dfsph_nx190_strict"]
    got = bench.read_traffic("dfsph_nx190_strict")
    if entry["source_hash"] == h:
        assert got["hbm_bytes_per_launch"] == entry["hbm_bytes_per_launch"] and got["hbm_bytes_per_launch"] > 4.5e8   # more than the algorithmic 0.45 GB
        assert "valu_issue_frac" in got and (got["valu_issue_frac"] is None or 0.0 < got["valu_issue_frac"] <= 1.05), "calibrated, never clipped"
        assert "VALU issue" in bench.limiter_text(got)
    else:
        assert got is None                                                # stale evidence is not reported
        assert "not measured for this build" in bench.limiter_text(got)   # ... and no limiter is asserted for it
    assert bench.read_traffic("no_such_workload") is None
    # SURVEY 8d's figures; fixed-count DFSPH leaves out the 44-byte error sweep nobody reads (DFSPHSolver::step)
    assert bench.step_bytes_per_particle("dfsph", 1, 4, 0, fixed=False) == 1000 and bench.step_bytes_per_particle("dfsph", 1, 4, 0) == 956
    assert bench.step_bytes_per_particle("pbd", 0, 0, 4) == 788


def test_tuning_block_round_trip_and_validation(sphx):
    """sphx_tuning (r05: the library's behaviour switches, formerly SPHX_* environment variables): defaults, set / get, refusal of a
    block built against another layout or with fields out of range, and the translation the test suite itself relies on
    (tests/tuning_env.py: variables present in the environment -> fields).  Host-only, no GPU."""
    import ctypes as C
    import tuning_env
    d = sphx.default_tuning()
    assert d.struct_size == C.sizeof(sphx.Tuning) and d.row_capacity == 0 and d.quad_mask == -1 and d.dfsph_window == -1 and d.pbd_skin < 0
    try:
        sphx.set_tuning(row_capacity=64, dfsph_no_tail=1, pbd_skin=0.1)
        g = sphx.get_tuning()
        assert (g.row_capacity, g.dfsph_no_tail) == (64, 1) and abs(g.pbd_skin - 0.1) < 1e-7
        bad = sphx.default_tuning(); bad.struct_size -= 4
        with pytest.raises(sphx.SphxError):
            sphx.set_tuning(bad)
        with pytest.raises(sphx.SphxError):
            sphx.set_tuning(row_capacity=5)
        with pytest.raises(sphx.SphxError):
            sphx.set_tuning(no_such_field=1)
        assert sphx.get_tuning().row_capacity == 64, "a refused block changes nothing"
        t = tuning_env.from_environment(sphx, {"SPHX_NBR_CAP": "12", "SPHX_DFSPH_NO_TAIL": "1", "SPHX_COMM_PRIORITY": "low",
                                               "SPHX_SLAB_EDGE_STREAM": "0", "SPHX_PBD_SKIN": "0.3", "SPHX_DFSPH_WINDOW": "0"})
        assert (t.row_capacity, t.dfsph_no_tail, t.slab_comm_priority, t.slab_edge_stream, t.dfsph_window) == (12, 1, 2, 0, 0)
        assert abs(t.pbd_skin - 0.3) < 1e-7
    finally:
        sphx.set_tuning()
    assert sphx.get_tuning().row_capacity == 0


def test_library_reads_no_behaviour_switch_from_the_environment():
    """getenv survives in the product sources only for the RCCL library path and, inside #ifdef SPHX_TEST_HOOKS, for
    the test build's fault injection."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "cpp-fluid-particles_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        hooks = 0
        for n, line in enumerate(open(path), 1):
            if line.lstrip().startswith("#ifdef SPHX_TEST_HOOKS"):
                hooks += 1
            elif line.lstrip().startswith("#endif") and hooks:
                hooks -= 1
            if "getenv(" in line.split("//")[0] and not hooks:
                assert "SPHX_RCCL_LIBRARY" in line, "%s:%d reads the environment: %s" % (path, n, line.strip())
