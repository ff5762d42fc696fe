"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on identical inputs.

Bar: bit-exact for integer outputs (cell ids, cell starts, permutation/ids) AND for every fp32
field — the engine keeps the reference's IEEE operation order, so equality is exact, not 1e-5.
The stated north-star tolerance (1e-5 relative) is asserted as well where a looser check is the
right statement (full-size properties)."""
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, same_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

FIELDS_COMMON = ["POS", "VEL", "DENSITY", "PRESSURE", "MASS", "CELL", "CELLSTART_F", "CELLSTART_B", "ID",
                 "BPOS", "BMASS"]
FIELDS_DFSPH = ["ALPHA", "KAPPA", "ERROR", "WARM"]
FIELDS_PBD = ["POS_LAST", "LAMBDA"]


def compare(sphx, oracle, gs, os_, names, tag):
    for nm in names:
        assert_bit_equal(gs.get(getattr(sphx, "F_" + nm)), os_.get(getattr(oracle, "F_" + nm)), "%s %s" % (tag, nm))


def make_pair(sphx, oracle, nx, solver, ctor_step=True, tweak=None):
    P, fluid, boundary = sphx.scene(nx)
    P.solver = solver
    if tweak:
        tweak(P)
    Po = same_params(oracle.Params(), P)
    return sphx.System(P, fluid, boundary, ctor_step), oracle.System(Po, fluid, boundary, ctor_step), P


def test_device_ieee_ops_match_host(sphx):
    """division, sqrt, float->int truncation and the uncontracted a*b+c are bit-identical to x86."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 8, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 8, n)).astype(np.float32)
    b[b == 0] = 1.0
    c = (rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 8, n)).astype(np.float32)
    # realistic ranges of the hot path as well: positions / cell length
    a[: n // 4] = rng.uniform(0, 4, n // 4).astype(np.float32)
    b[: n // 4] = np.float32(0.0404)
    q, r, t, m = sphx.ieee_probe(a, b, c)
    with np.errstate(all="ignore"):
        assert_bit_equal(q, a / b, "a/b")
        assert_bit_equal(r, np.sqrt(np.abs(a)), "sqrt")
        assert_bit_equal(m, (a * b).astype(np.float32) + c, "a*b+c (no FMA)")
        quo = a / b
        ok = np.abs(quo) < 2.0e9
        assert np.array_equal(t[ok], quo[ok].astype(np.int32)), "trunc"


def test_smoothing_kernels_pointwise(sphx, oracle):
    rng = np.random.default_rng(11)
    R = np.float32(0.04)
    r3 = rng.uniform(-1.3 * R, 1.3 * R, (200000, 3)).astype(np.float32)
    # exact edge cases: zero, on-support, half support, just inside/outside
    edge = np.array([[0, 0, 0], [R, 0, 0], [0, R / 2, 0], [0, 0, np.nextafter(R, np.float32(1))],
                     [np.nextafter(R, np.float32(0)), 0, 0], [1e-7, 0, 0], [1e-9, 1e-9, 0]], np.float32)
    r3 = np.concatenate([edge, r3])
    got = sphx.eval_kernels(r3, float(R))
    want = oracle.eval_kernels(r3, float(R))
    for g, w, nm in zip(got, want, ["W", "gradW", "viscLap", "surfGrad"]):
        assert_bit_equal(g, w, nm)


@pytest.mark.parametrize("nx", [8, 24])
def test_grid_and_boundary_mass_bit_exact(sphx, oracle, nx):
    gs, os_, _ = make_pair(sphx, oracle, nx, sphx.WCSPH, ctor_step=False)
    compare(sphx, oracle, gs, os_, ["CELL", "CELLSTART_F", "CELLSTART_B", "ID", "POS", "VEL", "BPOS", "BMASS", "MASS"], "init nx=%d" % nx)
    cs = gs.get(sphx.F_CELLSTART_F)
    assert cs[0] == 0 and np.all(np.diff(cs) >= 0) and cs[-1] <= gs.n


@pytest.mark.parametrize("solver,steps,dt", [(0, 12, 0.001), (1, 8, 0.002), (2, 6, 0.002)])
def test_trajectory_bit_exact_reference_scene(sphx, oracle, solver, steps, dt):
    """the reference's own scene (20,736 + 14,408 particles), ctor step + N steps, all fields."""
    def tweak(P):
        P.dt = dt
        P.pbd_iters = 5
    gs, os_, _ = make_pair(sphx, oracle, 24, solver, tweak=tweak)
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    compare(sphx, oracle, gs, os_, names, "ctor")
    for s in range(steps):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "solver %d step %d" % (solver, s + 1))
        if solver == 1:
            assert gs.iters() == os_.iters()


def _splash_state(n, P, seed):
    rng = np.random.default_rng(seed)
    lo = 0.02 * P.space[0]
    pos = rng.uniform(lo, 0.5 * P.space[0], (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(lo, 0.35 * P.space[1], n).astype(np.float32)
    vel = rng.normal(0, 0.8, (n, 3)).astype(np.float32)
    return pos, vel


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_trajectory_bit_exact_disordered(sphx, oracle, solver):
    """irregular neighbourhoods: uniformly random positions (ragged cells, many wall contacts,
    clamping, particles crossing cells every step) and random velocities; adaptive DFSPH."""
    def tweak(P):
        P.pbd_iters = 4
        P.dt = 0.001
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver; tweak(P)
    n = len(fluid)
    pos, vel = _splash_state(n, P, 5 + solver)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    # after the constructor's sort the arrays are permuted; set velocities in that order
    ids = gs.get(sphx.F_ID)
    assert_bit_equal(ids, os_.get(oracle.F_ID), "ids")
    gs.set(sphx.F_VEL, vel[ids]); os_.set(oracle.F_VEL, vel[ids])
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    for s in range(6):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "disordered solver %d step %d" % (solver, s + 1))
        if solver == 1:
            assert gs.iters() == os_.iters()


@pytest.mark.parametrize("solver", [0, 1, 2])
@pytest.mark.parametrize("flags,cap", [(1, None), (2, None), (4, None), (5, None), (0, "8"), (4, "8"), (16, None), (0, "quad-all"), (0, "quad-all-8"), (0, "duo-all"), (0, "duo-all-8"),
                                       (0, "lane-builder"), (0, "lane-builder-8")])
def test_engine_schedules_agree(sphx, oracle, solver, flags, cap, monkeypatch):
    """the fused/unfused schedules (bit 0), the neighbour-list vs direct 27-cell walks (bit 1), the
    LDS-staged tiles vs global gathers (bit 2 = on), the per-lane overflow fallback of the list (tiny
    capacity), lane-per-particle walks only (bit 4), quad-per-particle walks in EVERY sweep that has the variant
    (SPHX_QUAD_MASK; the default switches it on for the DFSPH rate sweeps only) and two-lanes-per-particle walks
    (SPHX_DUO_MASK; off by default) all produce the oracle's bits.  Scenes of this size build their rows with 16 lanes per particle
    (r06); "lane-builder" switches that off: the lane-per-particle builder of large scenes on the same states."""
    if cap and cap.startswith("lane-builder"):
        monkeypatch.setenv("SPHX_GROUP_BUILD_MAX", "-1")
        cap = cap[13:] or None
    if cap and cap.startswith("quad-all"):
        monkeypatch.setenv("SPHX_QUAD_MASK", "255")
        cap = cap[9:] or None
    if cap and cap.startswith("duo-all"):        # two lanes per particle in every sweep that has the variant
        monkeypatch.setenv("SPHX_QUAD_MASK", "0")
        monkeypatch.setenv("SPHX_DUO_MASK", "255")
        cap = cap[8:] or None
    if cap:
        monkeypatch.setenv("SPHX_NBR_CAP", cap)
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver; P.pbd_iters = 3; P.dt = 0.001
    n = len(fluid)
    pos, vel = _splash_state(n, P, 40 + solver)
    Po = same_params(oracle.Params(), P)
    P.reserved[0] = flags
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    ids = gs.get(sphx.F_ID)
    gs.set(sphx.F_VEL, vel[ids]); os_.set(oracle.F_VEL, vel[ids])
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    for s in range(4):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "flags %d cap %s solver %d step %d" % (flags, cap, solver, s + 1))


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_tile_schedule_is_only_a_schedule(sphx, oracle, solver, monkeypatch):
    """the (y-chunk, x) launch order of the 64-particle tiles (normally enabled for large scenes
    only) is forced on: results must not change"""
    monkeypatch.setenv("SPHX_FORCE_TILE_ORDER", "1")
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver; P.pbd_iters = 3; P.dt = 0.001
    pos, vel = _splash_state(len(fluid), P, 70 + solver)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    ids = gs.get(sphx.F_ID)
    gs.set(sphx.F_VEL, vel[ids]); os_.set(oracle.F_VEL, vel[ids])
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    for s in range(4):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "tile schedule solver %d step %d" % (solver, s + 1))


@pytest.mark.parametrize("skin,speed", [("0", 1.0), ("0.002", 1.0), ("0.05", 1.0), ("0.4", 1.0)])
def test_pbd_skin_rows_are_exact(sphx, oracle, skin, speed, monkeypatch):
    """PBD builds its neighbour rows once per step with an enlarged cutoff (skin) and re-tests every pair per sweep.
    The position update of every Jacobi iteration watches, on the device, whether a particle has moved farther than
    the skin allows or has left the cell its row was built around (the reference walks the 27 cells around the cell of
    the CURRENT position, PBDSolver.cu:139-141); if so the next iteration starts with a rebuild, otherwise the rows are
    kept.  skin 0 = an unconditional rebuild per iteration (the r01 behaviour).  A violent splash (speed 1) rebuilds in
    practically every iteration; a slow one (speed 0.02) keeps rows across iterations.  All equal the oracle bit for bit."""
    monkeypatch.setenv("SPHX_PBD_SKIN", skin)
    monkeypatch.setenv("SPHX_PBD_SKIN_FIXED", "1")          # no controller: this test pins the mode
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.PBD; P.pbd_iters = 5; P.dt = 0.002
    pos, vel = _splash_state(len(fluid), P, 333)
    vel = (vel * np.float32(speed)).astype(np.float32)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    gs.step(); os_.step()                        # records the positions (PBDSolver.cu:45-49)
    ids = gs.get(sphx.F_ID)
    last = (pos - np.float32(P.dt) * vel).astype(np.float32)[ids]     # velocities enter through the last positions
    gs.set(sphx.F_POS_LAST, last); os_.set(oracle.F_POS_LAST, last)
    for s in range(5):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, FIELDS_COMMON + FIELDS_PBD, "skin %s step %d" % (skin, s + 1))
    rebuilds_eager = gs.rows_stale()
    gs.step_n(3)
    for _ in range(3):
        os_.step()
    compare(sphx, oracle, gs, os_, FIELDS_COMMON + FIELDS_PBD, "skin %s graph replay" % skin)
    assert rebuilds_eager >= 0


@pytest.mark.parametrize("skin,partial", [("0.05", True), ("0.3", True), ("0.05", False), ("0.3", False)])
def test_pbd_skin_rows_kept_across_iterations(sphx, oracle, skin, partial, monkeypatch):
    """the dam-break column landing gently (nx = 12: contact after ~50 steps): particles move a little in every Jacobi
    iteration, mostly inside their cells, so rows are re-used across iterations and only sometimes rebuilt -- as a whole when a
    particle has moved farther than the skin allows, row by row for particles that changed their cell (r06; `partial` off: as a
    whole for those too, the r03-r05 rule)"""
    monkeypatch.setenv("SPHX_PBD_SKIN", skin)
    monkeypatch.setenv("SPHX_PBD_SKIN_FIXED", "1")
    if not partial:
        monkeypatch.setenv("SPHX_PBD_NO_PARTIAL", "1")
    def tweak(P):
        P.pbd_iters = 4
    gs, os_, P = make_pair(sphx, oracle, 12, sphx.PBD, tweak=tweak)
    iterations = 0
    for s in range(1, 91):
        gs.step(); os_.step()
        iterations += 4
        if s % 10 == 0 or s > 80:
            compare(sphx, oracle, gs, os_, FIELDS_COMMON + FIELDS_PBD, "gentle landing skin %s step %d" % (skin, s))
    rebuilds, rowwise = gs.rows_stale(), gs.rows_partial()
    if partial:
        assert rowwise > 0, "no particle changed its cell inside a step?"
        assert rebuilds < iterations, "every iteration rebuilt all rows: %d of %d" % (rebuilds, iterations)
    else:
        assert rowwise == 0
        assert 0 < rebuilds < iterations, "expected some, not all, iterations to rebuild: %d of %d" % (rebuilds, iterations)


def test_pbd_skin_controller_switches_modes_exactly(sphx, oracle):
    """the host-side controller reads the device-side rebuild counter every 32 steps and turns skin rows off while
    most iterations rebuild anyway (violent phase), back on later; whatever it decides, results equal the oracle"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.PBD; P.pbd_iters = 3; P.dt = 0.002
    pos, vel = _splash_state(len(fluid), P, 91)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    gs.step(); os_.step()
    ids = gs.get(sphx.F_ID)
    last = (pos - np.float32(P.dt) * vel).astype(np.float32)[ids]
    gs.set(sphx.F_POS_LAST, last); os_.set(oracle.F_POS_LAST, last)
    for block in range(5):                      # 5 x 40 steps: crosses several controller periods, eager and replayed
        if block % 2 == 0:
            for _ in range(40):
                gs.step()
        else:
            gs.step_n(40)
        for _ in range(40):
            os_.step()
        compare(sphx, oracle, gs, os_, ["POS", "VEL", "DENSITY", "ID", "POS_LAST"], "controller block %d" % block)


def test_dfsph_fixed_iterations_and_graph_replay(sphx, oracle):
    """fixed (v=1, d=4) mode: step_n replays a captured hipGraph; results equal eager stepping."""
    def tweak(P):
        P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
    gs, os_, P = make_pair(sphx, oracle, 24, sphx.DFSPH, tweak=tweak)
    gs.step_n(7)
    for _ in range(7):
        os_.step()
    compare(sphx, oracle, gs, os_, FIELDS_COMMON + FIELDS_DFSPH, "graph replay")
    assert gs.iters() == (1, 4)


@pytest.mark.parametrize("solver", [0, 2])
def test_step_n_matches_step(sphx, oracle, solver):
    def tweak(P):
        P.pbd_iters = 3
    gs, os_, P = make_pair(sphx, oracle, 12, solver, tweak=tweak)
    gs.step()
    os_.step()      # PBD: first real step after the "throw" step of the constructor
    gs.step_n(5)
    for _ in range(5):
        os_.step()
    compare(sphx, oracle, gs, os_, FIELDS_COMMON, "step_n solver %d" % solver)


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_row_builders_agree_below_their_crossover(sphx, solver, monkeypatch):
    """69,984 particles (just under sphx_tuning.group_build_max, where the 16-lanes-per-particle builder is still the default): the
    same disordered state stepped with that builder and with the lane-per-particle builder of large scenes -- every field, the row
    statistics and the iteration counts are identical, bit for bit (engine against engine; the oracle pins both at 2,592 and 20,736
    particles in the tests above)"""
    P, fluid, boundary = sphx.scene(36)
    P.solver = solver; P.pbd_iters = 4; P.dt = 0.001
    pos, vel = _splash_state(len(fluid), P, 300 + solver)
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    results = []
    for group in (True, False):
        monkeypatch.setenv("SPHX_GROUP_BUILD_MAX", "0" if group else "-1")
        s = sphx.System(P, pos, boundary, ctor_step=False)
        ids = s.get(sphx.F_ID)
        s.set(sphx.F_VEL, vel[ids])
        for _ in range(3):
            s.step()
        s.step_n(3)
        results.append(({n: s.get(getattr(sphx, "F_" + n)) for n in names}, s.row_stats()[:2], s.iters() if solver == 1 else None))
        s.close()
    (a, ra, ia), (b, rb, ib) = results
    for n in names:
        assert_bit_equal(a[n], b[n], "builders, solver %d, %s" % (solver, n))
    assert ra == rb and ia == ib


def test_out_of_grid_ranks_across_many_scan_tiles(sphx, oracle):
    """the stable ranks of the out-of-grid bucket come from ONE guarded launch (r06: decoupled look-back over tiles of 2,048 particles,
    csrc/scan_chain.hpp): 263,424 particles = 129 tiles (three look-back windows), every 11th particle far outside the box, so that every
    tile holds some; the sort (ids, cells, cell table) and the first steps equal the oracle's.  The cell table of this scene (216,001
    words = 106 tiles) goes through the one-launch scan too."""
    P, fluid, boundary = sphx.scene(56)
    P.solver = sphx.WCSPH
    pos = fluid.copy()
    pos[::11] += np.float32(7.0)
    pos[3::5000] = [-0.3, 0.1, 0.1]
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    names = ["POS", "VEL", "DENSITY", "PRESSURE", "CELL", "CELLSTART_F", "ID"]
    compare(sphx, oracle, gs, os_, names, "many tiles init")
    for s in range(2):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "many tiles step %d" % (s + 1))
    gs.step_n(3)
    for _ in range(3):
        os_.step()
    compare(sphx, oracle, gs, os_, names, "many tiles, replayed steps")


def test_edge_cases_no_boundary_and_out_of_grid(sphx, oracle):
    """no boundary particles at all; some particles outside the grid (sentinel cell)."""
    P, fluid, boundary = sphx.scene(8)
    P.solver = sphx.WCSPH
    pos = fluid.copy()
    pos[::37] += np.float32(5.0)        # far outside the box -> sentinel cell until clamped
    pos[5] = [-0.3, 0.1, 0.1]
    Po = same_params(oracle.Params(), P)
    empty = np.zeros((0, 3), np.float32)
    gs = sphx.System(P, pos, empty, ctor_step=False)
    os_ = oracle.System(Po, pos, empty, ctor_step=False)
    names = ["POS", "VEL", "DENSITY", "PRESSURE", "CELL", "CELLSTART_F", "CELLSTART_B", "ID"]
    compare(sphx, oracle, gs, os_, names, "edge init")
    for s in range(3):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "edge step %d" % s)


@pytest.mark.parametrize("solver", [0, 1])
def test_static_obstacles_boundary_mass_and_trajectory(sphx, oracle, solver):
    """§8f-4: a box, a sphere and a triangle-soup ramp sampled into boundary particles and appended to the shell;
    the boundary grid, the boundary masses (SPHSystem.cu:79-112 on a non-shell set) and the fluid trajectory
    falling onto them equal the oracle bit for bit"""
    P, fluid, shell = sphx.scene(12)
    P.solver = solver
    P.dt = 0.001 if solver == 0 else 0.002
    box = sphx.sample_box((0.40, 0.0, 0.10), (0.46, 0.15, 0.40), 0.02)
    ball = sphx.sample_sphere((0.25, 0.02, 0.25), 0.018, 0.01)
    ramp = sphx.sample_triangles(np.float32([[0.05, 0.0, 0.05, 0.13, 0.0, 0.05, 0.05, 0.06, 0.45],
                                             [0.13, 0.0, 0.05, 0.13, 0.06, 0.45, 0.05, 0.06, 0.45]]), 0.02)
    boundary = np.concatenate([shell, box, ball, ramp]).astype(np.float32)
    assert len(boundary) > len(shell) + 400
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, fluid, boundary, ctor_step=False)
    os_ = oracle.System(Po, fluid, boundary, ctor_step=False)
    compare(sphx, oracle, gs, os_, ["CELLSTART_B", "BPOS", "BMASS", "CELL", "CELLSTART_F", "ID"], "obstacle init")
    bm = gs.get(sphx.F_BMASS)
    assert np.isfinite(bm).all() and bm.min() > 0
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else [])
    vel = np.zeros_like(fluid); vel[:, 1] = -1.5; vel[:, 0] = 0.8          # drive the block into the obstacles
    ids = gs.get(sphx.F_ID)
    gs.set(sphx.F_VEL, vel[ids]); os_.set(oracle.F_VEL, vel[ids])
    touched = False
    for s in range(12):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "obstacles solver %d step %d" % (solver, s + 1))
    # the fluid did meet the obstacle particles: some rows contain boundary neighbours beyond the shell's reach
    pos = gs.get(sphx.F_POS)
    d = np.linalg.norm(pos[:, None, :] - ball[None, ::4, :], axis=2).min()
    assert d < 0.04, "the block must come within the support radius of the sphere obstacle"


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_trajectory_bit_exact_other_constants(sphx, oracle, solver):
    """every scalar away from the reference scene's values: rest density != 1 (the divisions by rho0 that the scene's
    rho0 = 1 lets the engine skip), another support radius and cell length (the validated fast division / sqrt paths are
    per radius), other boundary density, mass, stiffness, viscosity, surface coefficients, tilted gravity"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver
    P.radius = 0.05; P.cell_length = 1.03 * P.radius
    for a in range(3):
        P.cells[a] = int(np.ceil(P.space[a] / P.cell_length))
    P.rho0 = 1.3; P.rho_boundary = 1.9; P.m0 = 9.1e-5; P.stiff = 15.0; P.visc = 1.1e-3
    P.surface_tension = 2.3e-4; P.air_pressure = 3.1e-4
    P.gravity[0] = 1.0; P.gravity[1] = -9.0; P.gravity[2] = 0.5
    P.dt = 0.001; P.pbd_iters = 4; P.pbd_xsph_c = 0.07; P.pbd_relaxation = 0.6
    pos, vel = _splash_state(len(fluid), P, 55 + solver)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    ids = gs.get(sphx.F_ID)
    assert_bit_equal(ids, os_.get(oracle.F_ID), "ids")
    gs.set(sphx.F_VEL, vel[ids]); os_.set(oracle.F_VEL, vel[ids])
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    for s in range(6):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, names, "other constants solver %d step %d" % (solver, s + 1))
        if solver == 1:
            assert gs.iters() == os_.iters()
    gs.step_n(3)
    for _ in range(3):
        os_.step()
    compare(sphx, oracle, gs, os_, names, "other constants solver %d graph" % solver)


def test_single_particle(sphx, oracle):
    P, fluid, boundary = sphx.scene(8)
    P.solver = sphx.DFSPH
    Po = same_params(oracle.Params(), P)
    one = fluid[:1].copy()
    gs = sphx.System(P, one, boundary); os_ = oracle.System(Po, one, boundary)
    for _ in range(3):
        gs.step(); os_.step()
    compare(sphx, oracle, gs, os_, FIELDS_COMMON + FIELDS_DFSPH, "single particle")


def test_golden_fixture(sphx):
    """committed golden vectors (tests/golden, generated by tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dambreak_nx8.npz"))
    for key, solver in (("wcsph", 0), ("dfsph", 1), ("pbd", 2)):
        P, fluid, boundary = sphx.scene(8)
        P.solver = solver
        P.pbd_iters = 4
        s = sphx.System(P, fluid, boundary)
        for _ in range(int(g["steps"])):
            s.step()
        assert_bit_equal(s.get(sphx.F_POS), g[key + "_pos"], key + " golden pos")
        assert_bit_equal(s.get(sphx.F_DENSITY), g[key + "_density"], key + " golden density")
        assert np.array_equal(s.get(sphx.F_CELL), g[key + "_cell"])


def test_full_size_properties_dfsph_1m(sphx):
    """BASELINE config 3 (1,022,208 particles, DFSPH v=1,d=4): size-independent properties."""
    P, fluid, boundary = sphx.scene(88)
    P.solver = sphx.DFSPH
    P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
    s = sphx.System(P, fluid, boundary)
    n = s.n
    assert n == 1022208
    s.step_n(3)
    ids = s.get(sphx.F_ID)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.int32)), "ids must stay a permutation"
    cs = s.get(sphx.F_CELLSTART_F)
    assert cs[0] == 0 and np.all(np.diff(cs) >= 0) and cs[-1] == n
    pos = s.get(sphx.F_POS); den = s.get(sphx.F_DENSITY); vel = s.get(sphx.F_VEL)
    assert np.isfinite(pos).all() and np.isfinite(den).all() and np.isfinite(vel).all()
    assert pos.min() >= 0 and pos.max() <= 0.99 * P.space[0] + 1e-6
    # free fall: after k steps (ctor + 3) every interior particle has v_y = -(k) * g * dt up to the
    # solver's corrections; the mean must be close, the lattice density close to the 20k scene's
    assert abs(vel[:, 1].mean() + 4 * 9.8 * P.dt) < 0.2 * 4 * 9.8 * P.dt
    assert 0.70 < den.mean() < 0.85
    # idempotence of the neighbour search: cells computed from the sorted positions are sorted
    cell = s.get(sphx.F_CELL)
    s.step()
    cell2 = s.get(sphx.F_CELL)   # keys in pre-sort order of this step = sorted order of last step
    moved = np.count_nonzero(np.diff(cell2) < 0)
    assert moved < n // 100
    del cell


def test_cpp_api_driver_matches_oracle(oracle, tmp_path):
    """apps/sphx_demo is a main.cpp-style driver compiled against include/*.h (the C++ drop-in API,
    default adaptive DFSPH exactly as main.cpp constructs it); its dump must equal the oracle."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "sphx_demo")
    subprocess.check_call(["make", "-C", os.path.join(root, "apps")], stdout=subprocess.DEVNULL)   # no-op when up to date
    out = str(tmp_path / "dump.bin")
    subprocess.check_call([exe, "--solver", "dfsph", "--nx", "12", "--steps", "6", "--dump", out])
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    pos = np.frombuffer(raw[4:4 + 12 * n], np.float32).reshape(n, 3)
    den = np.frombuffer(raw[4 + 12 * n:], np.float32)
    P, fluid, boundary = oracle.scene(12)
    P.solver = oracle.DFSPH
    o = oracle.System(P, fluid, boundary)
    for _ in range(6):
        o.step()
    assert_bit_equal(pos, o.get(oracle.F_POS), "demo pos")
    assert_bit_equal(den, o.get(oracle.F_DENSITY), "demo density")


@pytest.mark.parametrize("radius", [0.04, 0.013, 0.5])
def test_exact_fast_paths_match_ieee_operators(sphx, radius):
    """the rcp/sqrt + FMA-refinement fast paths equal the plain IEEE operators: x/R exhaustively over
    every float in [0, 2.2R], sqrt exhaustively over every non-negative finite float, the shared-
    denominator division over 2^28 pseudo-random triples incl. zero and tiny numerators."""
    bad, enabled = sphx.fastmath_selftest(radius)
    assert bad[1] == 0, "sqrt_exact differs from sqrtf"
    assert bad[2] == 0, "div3_exact differs from IEEE division"
    assert enabled[0] == (1 if bad[0] == 0 else 0)
    if radius == 0.04:
        assert enabled == [1, 1], "the reference radius must run on the fast paths"


@pytest.mark.parametrize("solver", [0, 1, 2])
def test_snapshot_resume_matches_oracle(sphx, oracle, tmp_path, solver):
    """save after k steps of a disordered splash (particles change cells every step, so the arrays are
    NOT in sorted order when saved), reload through the C ABI, continue: equals the ORACLE's
    uninterrupted run bit for bit (incl. DFSPH warm start, adaptive iteration counts, PBD last positions)"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver; P.pbd_iters = 3; P.dt = 0.001
    pos, vel = _splash_state(len(fluid), P, 90 + solver)
    Po = same_params(oracle.Params(), P)
    a = sphx.System(P, pos, boundary, ctor_step=False)
    o = oracle.System(Po, pos, boundary, ctor_step=False)
    ids = a.get(sphx.F_ID)
    a.set(sphx.F_VEL, vel[ids]); o.set(oracle.F_VEL, vel[ids])
    for _ in range(4):
        a.step(); o.step()
    assert np.count_nonzero(np.diff(a.get(sphx.F_CELL)) < 0) > 0, "the saved state must be out of cell order"
    path = str(tmp_path / "snap.bin")
    sphx.save_snapshot(a, path)
    saved = {f: a.get(f) for f in (sphx.F_POS, sphx.F_VEL, sphx.F_ID, sphx.F_DENSITY)}
    a.close()
    b = sphx.load_snapshot(path)
    assert b.n == len(fluid) and b.params.solver == solver
    for f, v in saved.items():
        assert_bit_equal(b.get(f), v, "restored field %d" % f)
    names = ["POS", "VEL", "DENSITY", "PRESSURE", "ID", "CELL", "CELLSTART_F"] + (FIELDS_DFSPH if solver == 1 else []) + \
            (FIELDS_PBD if solver == 2 else [])
    for s_ in range(3):
        b.step(); o.step()
        compare(sphx, oracle, b, o, names, "resumed solver %d step %d" % (solver, s_ + 1))
        if solver == 1:
            assert b.iters() == o.iters()


def _size_independent_checks(sphx, s, P, n_expect):
    n = s.n
    assert n == n_expect
    ids = s.get(sphx.F_ID)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.int32)), "ids must stay a permutation"
    cs = s.get(sphx.F_CELLSTART_F)
    assert cs[0] == 0 and np.all(np.diff(cs) >= 0) and cs[-1] == n
    pos = s.get(sphx.F_POS); den = s.get(sphx.F_DENSITY); vel = s.get(sphx.F_VEL)
    assert np.isfinite(pos).all() and np.isfinite(den).all() and np.isfinite(vel).all()
    assert pos.min() >= 0 and pos.max() <= 0.99 * P.space[0] + 1e-6
    return pos, vel, den


def test_full_size_properties_wcsph_263k(sphx):
    """BASELINE config 2 (263,424 particles, WCSPH, dt = 0.001): size-independent properties."""
    P, fluid, boundary = sphx.scene(56)
    P.solver = sphx.WCSPH; P.dt = 0.001
    s = sphx.System(P, fluid, boundary)
    s.step_n(5)
    pos, vel, den = _size_independent_checks(sphx, s, P, 263424)
    k = 6                                     # ctor step + 5
    assert abs(vel[:, 1].mean() + k * 9.8 * P.dt) < 0.25 * k * 9.8 * P.dt      # free fall dominates
    assert 0.70 < den.mean() < 0.85 and den.max() < 1.3
    pr = s.get(sphx.F_PRESSURE)
    assert pr.min() >= 0.0                     # Tait pressure is clamped at zero (BasicSPHSolver.cu:110)
    # the lattice is symmetric under x <-> z: mean displacement in x and z must agree to rounding
    disp = pos[np.argsort(s.get(sphx.F_ID))] - fluid
    assert abs(disp[:, 0].mean() - disp[:, 2].mean()) < 1e-6


def test_full_size_properties_pbd_1m(sphx):
    """BASELINE config 4 (1,022,208 particles, PBD, 4 Jacobi iterations + XSPH): properties."""
    P, fluid, boundary = sphx.scene(88)
    P.solver = sphx.PBD; P.pbd_iters = 4
    s = sphx.System(P, fluid, boundary)        # the constructor step only records positions (PBDSolver.cu:45-49)
    assert np.all(s.get(sphx.F_DENSITY) == 0.0)
    s.step_n(4)
    pos, vel, den = _size_independent_checks(sphx, s, P, 1022208)
    assert 0.70 < den.mean() < 0.90
    # PBD velocity = displacement / dt of the last step (PBDSolver.cu:55-60) + XSPH + surface + gravity:
    # the mean fall speed after 4 real steps is 4 g dt to within the constraint corrections
    assert abs(vel[:, 1].mean() + 4 * 9.8 * P.dt) < 0.3 * 4 * 9.8 * P.dt
    last = s.get(sphx.F_POS_LAST)
    assert np.isfinite(last).all() and np.abs(pos - last).max() < 0.05


def test_10m_path_smoke(sphx):
    """BASELINE config 5's particle count on one device (10,288,500 particles, DFSPH v=1, d=4): the
    index arithmetic above 2^31 bytes, the automatic (y-chunk, x) tile schedule and the row storage at
    this size are exercised by a test, not only by bench.py"""
    P, fluid, boundary = sphx.scene(190)
    P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
    s = sphx.System(P, fluid, boundary)
    s.step_n(2)
    pos, vel, den = _size_independent_checks(sphx, s, P, 10288500)
    assert abs(vel[:, 1].mean() + 3 * 9.8 * P.dt) < 0.2 * 3 * 9.8 * P.dt
    assert 0.70 < den.mean() < 0.85
    # interior lattice particles all have the same neighbourhood: their densities agree to rounding noise
    mid = (np.abs(pos[:, 0] - 0.5 * P.space[0]) < 0.3) & (np.abs(pos[:, 2] - 0.5 * P.space[2]) < 0.3) & \
          (np.abs(pos[:, 1] - 0.4 * P.space[1]) < 0.3)
    assert mid.sum() > 1000 and np.ptp(den[mid]) < 1e-4
    s.close()


def test_rows_beyond_32bit_entry_count(sphx, monkeypatch):
    """49,152,000 particles (nx = 320) with the row capacity pinned at 96: cap x particles = 4.7e9 row entries, past a
    32-bit element count (the row store has a 64-bit length; round 1 fell back to direct walks beyond 44.7 M particles).
    Two WCSPH steps through the rows equal two steps of direct 27-cell walks (engine flag 2, oracle-checked at small
    sizes) bit for bit."""
    monkeypatch.setenv("SPHX_NBR_CAP", "96")
    P, fluid, boundary = sphx.scene(320)
    n = 320 * 480 * 320
    assert len(fluid) == n and ((n + 63) // 64) * 64 * 96 > 2 ** 32
    P.solver = sphx.WCSPH; P.dt = 0.001
    out = []
    for flags in (0, 2):
        Q = P.copy(); Q.reserved[0] = flags
        s = sphx.System(Q, fluid, boundary)
        s.step()
        if flags == 0:
            pairs, longest, _ = s.row_stats()
            assert pairs > 20 * n and 0 < longest <= 96, "the sweeps must have walked rows"
            _size_independent_checks(sphx, s, P, n)
        out.append([s.get(f) for f in (sphx.F_POS, sphx.F_DENSITY, sphx.F_ID)])
        s.close()
    for a, b, nm in zip(out[0], out[1], ["pos", "density", "id"]):
        assert_bit_equal(a, b, "49M rows vs direct walks " + nm)


@pytest.mark.parametrize("solver", [0, 1])
def test_raised_count_rows_fit_capacity(sphx, solver):
    """sphx_set_count may raise the active count after the neighbour rows were first built (slab
    drivers do every step): the row storage is sized for the capacity, so the result equals the
    direct 27-cell walk (engine flag 2 = no rows) bit for bit"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = solver; P.dt = 0.001
    if solver == 1:
        P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 2
    pos, vel = _splash_state(len(fluid), P, 120 + solver)
    out = []
    for flags in (0, 2):
        Q = P.copy(); Q.reserved[0] = flags
        s = sphx.System(Q, pos, boundary, ctor_step=False)
        ids = s.get(sphx.F_ID)
        s.set(sphx.F_VEL, vel[ids])
        s.set_count(s.n // 3)
        s.step()                      # rows first built for a third of the particles
        s.set_count(s.n)
        s.step(); s.step_n(2)
        out.append([s.get(f) for f in (sphx.F_POS, sphx.F_VEL, sphx.F_DENSITY, sphx.F_ID)])
        s.close()
    for a, b, nm in zip(out[0], out[1], ["pos", "vel", "density", "id"]):
        assert_bit_equal(a, b, "raised count " + nm)


def test_graph_follows_host_side_invalidation(sphx):
    """a captured hipGraph must not survive host-side changes: rewriting the boundary masses after
    step_n (snapshot restore does) has to reach the replayed steps exactly as it reaches eager ones;
    a changed active count re-captures as well"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 2
    a = sphx.System(P, fluid, boundary); b = sphx.System(P, fluid, boundary)
    a.step_n(3)
    for _ in range(3):
        b.step()
    bm = a.get(sphx.F_BMASS) * np.float32(1.5)
    a.set(sphx.F_BMASS, bm); b.set(sphx.F_BMASS, bm)
    a.step_n(3)
    for _ in range(3):
        b.step()
    for f in (sphx.F_POS, sphx.F_VEL, sphx.F_DENSITY):
        assert_bit_equal(a.get(f), b.get(f), "after BMASS rewrite, field %d" % f)
    a.set_count(a.n - 64); b.set_count(b.n - 64)
    a.step_n(2)
    b.step(); b.step()
    for f in (sphx.F_POS, sphx.F_VEL, sphx.F_DENSITY):
        assert_bit_equal(a.get(f)[: a.n - 64], b.get(f)[: b.n - 64], "after count change, field %d" % f)


def test_exceptions_do_not_cross_the_c_boundary(sphx):
    """C++ exceptions thrown inside the engine (`throw "text"` like the reference's PBD first step,
    PBDSolver.cu:45-49) come back as status codes with text, from every guarded entry point"""
    import ctypes as C
    P, fluid, boundary = sphx.scene(8)
    P.solver = sphx.WCSPH
    s = sphx.System(P, fluid, boundary, ctor_step=False)
    L = sphx.lib()
    rc = L.sphx_run_phase(s._h, sphx.PH_P_LAMBDA)          # a PBD stage on a WCSPH system
    assert rc == -4 and b"PBD" in L.sphx_last_error()
    rc = L.sphx_run_phase_reduce(s._h, sphx.PH_DEN_ERROR_ACC, 0, 10)
    assert rc == -4 and b"DFSPH" in L.sphx_last_error()
    tot = C.c_longlong()
    assert L.sphx_error_total_fixed(s._h, C.byref(tot)) == -4
    assert L.sphx_run_phase(s._h, 999) == -4
    s.step()                                                # the system is still usable
    assert np.isfinite(s.get(sphx.F_POS)).all()
    Q = P.copy(); Q.solver = 7
    h = C.c_void_p()
    assert L.sphx_create(C.byref(Q), fluid.ctypes.data, len(fluid), boundary.ctypes.data, len(boundary), 1, C.byref(h)) == -1
    assert L.sphx_snapshot_load(b"/nonexistent/snap.bin", C.byref(h)) == -1


def test_demo_solver_hot_swap_and_reference_generate_dots(oracle, tmp_path):
    """apps/sphx_demo --restart-with rebuilds the scene with another solver in the same process (the
    reference's keys '1' '2' '3', main.cpp:225-239): the second run must equal a fresh oracle run.
    --dots goes through the reference-signature `extern "C" generate_dots` (vbo.cu:46-51)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "sphx_demo")
    subprocess.check_call(["make", "-C", os.path.join(root, "apps")], stdout=subprocess.DEVNULL)
    out, dots = str(tmp_path / "dump.bin"), str(tmp_path / "dots.bin")
    subprocess.check_call([exe, "--solver", "pbd", "--restart-with", "dfsph", "--nx", "12", "--steps", "5", "--dump", out,
                           "--dots", dots], stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    pos = np.frombuffer(raw[4:4 + 12 * n], np.float32).reshape(n, 3)
    den = np.frombuffer(raw[4 + 12 * n:], np.float32)
    P, fluid, boundary = oracle.scene(12)
    P.solver = oracle.DFSPH
    o = oracle.System(P, fluid, boundary)
    for _ in range(5):
        o.step()
    assert_bit_equal(pos, o.get(oracle.F_POS), "hot-swapped run pos")
    assert_bit_equal(den, o.get(oracle.F_DENSITY), "hot-swapped run density")
    raw = open(dots, "rb").read()
    assert int(np.frombuffer(raw[:4], np.int32)[0]) == n
    dc = np.frombuffer(raw[4:], np.float32).reshape(2, n, 3)
    assert_bit_equal(dc[0], pos, "generate_dots positions")
    water, foam, dense = np.float32([0.34, 0.46, 0.7]), np.float32([0.9, 0.9, 0.9]), np.float32([1.0, 0.4, 0.7])
    w1 = np.clip((den - np.float32(0.75)) * np.float32(4.0), 0, 1)[:, None]
    w2 = np.minimum((den * den - np.float32(1.0)) * np.float32(4.0), np.float32(1.0))[:, None]
    want = np.where((den < 0.75)[:, None], water, np.where((den < 1.0)[:, None], w1 * foam + (1 - w1) * water,
                                                           (1 - w2) * foam + w2 * dense)).astype(np.float32)
    assert np.allclose(dc[1], want, rtol=0, atol=1e-6)


def test_demo_snapshot_continue(sphx, oracle, tmp_path):
    """the demo continues a snapshot (--load) and its dump equals the oracle's uninterrupted run"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "sphx_demo")
    subprocess.check_call(["make", "-C", os.path.join(root, "apps")], stdout=subprocess.DEVNULL)
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.DFSPH
    a = sphx.System(P, fluid, boundary)
    for _ in range(30):
        a.step()
    snap, out = str(tmp_path / "s.bin"), str(tmp_path / "d.bin")
    sphx.save_snapshot(a, snap)
    a.close()
    subprocess.check_call([exe, "--load", snap, "--steps", "4", "--dump", out], stdout=subprocess.DEVNULL)
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    Po = same_params(oracle.Params(), P)
    o = oracle.System(Po, fluid, boundary)
    for _ in range(34):
        o.step()
    assert_bit_equal(np.frombuffer(raw[4:4 + 12 * n], np.float32).reshape(n, 3), o.get(oracle.F_POS), "continued pos")


def test_generate_dots_colour_ramp(sphx):
    """generate_dots (vbo.cu:26-51): positions copied, density mapped to the reference colour ramp"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")            # the runtime instance libsphx.so is already bound to
    P, fluid, boundary = sphx.scene(8)
    s = sphx.System(P, fluid, boundary)
    n = s.n
    d_dot, d_col = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_dot), C.c_size_t(12 * n)) == 0 and hip.hipMalloc(C.byref(d_col), C.c_size_t(12 * n)) == 0
    assert sphx.lib().sphx_generate_dots(s._h, d_dot, d_col) == 0
    dot_h = np.empty((n, 3), np.float32); col_h = np.empty((n, 3), np.float32)
    assert hip.hipMemcpy(C.c_void_p(dot_h.ctypes.data), d_dot, C.c_size_t(12 * n), 2) == 0      # hipMemcpyDeviceToHost
    assert hip.hipMemcpy(C.c_void_p(col_h.ctypes.data), d_col, C.c_size_t(12 * n), 2) == 0
    hip.hipFree(d_dot); hip.hipFree(d_col)

    class _T:                                   # tiny shim so the checks below read the same
        def __init__(self, a): self.a = a
        def cpu(self): return self
        def numpy(self): return self.a
    dot, col = _T(dot_h), _T(col_h)
    assert_bit_equal(dot.cpu().numpy(), s.get(sphx.F_POS), "dots")
    rho = s.get(sphx.F_DENSITY).astype(np.float32)
    water, foam, dense = np.float32([0.34, 0.46, 0.7]), np.float32([0.9, 0.9, 0.9]), np.float32([1.0, 0.4, 0.7])
    want = np.empty((n, 3), np.float32)
    for i in range(n):
        r = rho[i]
        if r < 0.75:
            want[i] = water
        elif r < 1.0:
            w = (r - np.float32(0.75)) * np.float32(4.0)
            want[i] = w * foam + (np.float32(1) - w) * water
        else:
            w = min((r * r - np.float32(1.0)) * np.float32(4.0), np.float32(1.0))
            want[i] = (np.float32(1) - w) * foam + w * dense
    assert np.allclose(col.cpu().numpy(), want, rtol=0, atol=1e-6)


def test_generate_dots_every_branch_bit_exact(sphx):
    """generate_dots_CUDA (vbo.cu:26-44) with densities written into the device array so that every branch runs: below 0.75,
    the [0.75, 1) ramp, the [1, ...) ramp and its fminf clamp (rho^2 - 1 >= 0.25), the branch borders themselves, NaN
    (every comparison false: the last branch, where fminf(NaN, 1) = 1).  fp32 restatement, bit for bit."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    P, fluid, boundary = sphx.scene(8)
    s = sphx.System(P, fluid, boundary)
    n = s.n
    rng = np.random.default_rng(5)
    rho = rng.uniform(0.3, 1.4, n).astype(np.float32)
    special = np.float32([0.75, np.nextafter(np.float32(0.75), np.float32(0)), 1.0, np.nextafter(np.float32(1), np.float32(0)), np.nextafter(np.float32(1), np.float32(2)),
                          np.sqrt(np.float32(1.25)), 1.1180340, 1.12, 1.3, 0.0, -1.0, 1e6, np.nan, np.inf])
    rho[:len(special)] = special
    d_rho = C.c_void_p(s.device_ptr(sphx.F_DENSITY))
    assert hip.hipMemcpy(d_rho, C.c_void_p(rho.ctypes.data), C.c_size_t(4 * n), 1) == 0        # hipMemcpyHostToDevice
    d_dot, d_col = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_dot), C.c_size_t(12 * n)) == 0 and hip.hipMalloc(C.byref(d_col), C.c_size_t(12 * n)) == 0
    assert sphx.lib().sphx_generate_dots(s._h, d_dot, d_col) == 0
    dot_h = np.empty((n, 3), np.float32); col_h = np.empty((n, 3), np.float32)
    assert hip.hipMemcpy(C.c_void_p(dot_h.ctypes.data), d_dot, C.c_size_t(12 * n), 2) == 0
    assert hip.hipMemcpy(C.c_void_p(col_h.ctypes.data), d_col, C.c_size_t(12 * n), 2) == 0
    hip.hipFree(d_dot); hip.hipFree(d_col)
    assert_bit_equal(dot_h, s.get(sphx.F_POS), "dots")
    f32 = np.float32
    water, foam, dense = f32([0.34, 0.46, 0.7]), f32([0.9, 0.9, 0.9]), f32([1.0, 0.4, 0.7])
    want = np.empty((n, 3), np.float32)
    branches = [0, 0, 0, 0]
    with np.errstate(all="ignore"):
        for i in range(n):
            r = rho[i]
            if r < f32(0.75):
                want[i] = water; branches[0] += 1
            elif r < f32(1.0):
                w = f32(f32(r - f32(0.75)) * f32(4.0))
                want[i] = (w * foam).astype(np.float32) + (f32(f32(1) - w) * water).astype(np.float32); branches[1] += 1
            else:
                w = f32(f32(f32(r * r) - f32(1.0)) * f32(4.0))
                clamped = not (w < f32(1.0))                 # fminf(w, 1): 1 for w >= 1 and for NaN
                if clamped:
                    w = f32(1.0)
                want[i] = (f32(f32(1) - w) * foam).astype(np.float32) + (w * dense).astype(np.float32); branches[3 if clamped else 2] += 1
    assert min(branches) > 10, branches
    assert_bit_equal(col_h, want, "colours")
    s.close()


def test_particles_advect_public_method(tmp_path):
    """Particles::advect (Particles.cu:28-36) on its own, through the C++ API: pos += dt * vel, one multiply and one add per
    component in fp32 (no contraction), bit for bit"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "apps", "sphx_demo")
    subprocess.check_call(["make", "-C", os.path.join(root, "apps")], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "advect.bin")
    subprocess.check_call([exe, "--advect-check", out])
    raw = open(out, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0]); dt = np.frombuffer(raw[4:8], np.float32)[0]
    arr = np.frombuffer(raw[8:], np.float32).reshape(3, n, 3)
    want = (arr[0] + (dt * arr[1]).astype(np.float32)).astype(np.float32)
    assert n == 4099 and np.count_nonzero(arr[1]) > n
    assert_bit_equal(arr[2], want, "Particles::advect")


# ------------------------------------------------------------------------------------------------------------------
# Through the landing of the column on the reference scene: wall clamps, boundary contributions, the divergence-error
# clamp, adaptive DFSPH running into maxIter = 20.  (The short trajectory tests above end in free fall.)
def _anchor_states(name):
    import json, os
    V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refsrc_anchors.json")))["variants"]["float_fabs"][name]
    return V["dt"], {st["step"]: st for st in V["states"]}


def _crc_in_particle_order(sphx, gs, field):
    import zlib
    ids = gs.get(sphx.F_ID)
    a = gs.get(field); b = np.empty_like(a); b[ids] = a
    return zlib.crc32(b.tobytes())


def test_reference_source_anchor_dfsph_on_gpu(sphx):
    """The ENGINE (no oracle in the loop) against tests/golden/refsrc_anchors.json: CRC-32 of the complete pos / vel /
    density arrays and the iteration counts of the reference's own sources (compiled on CPU, see tests/golden/README.md) on
    the reference scene, adaptive DFSPH, every 10 steps up to step 100 = through the landing, iterations up to (20,2)."""
    dt, states = _anchor_states("dfsph")
    P, fluid, boundary = sphx.scene(24)
    P.solver = sphx.DFSPH; P.dt = dt
    gs = sphx.System(P, fluid, boundary)
    for step in range(0, 101):
        if step:
            gs.step()
        st = states.get(step)
        if st:
            for f, k in ((sphx.F_POS, "crc32_pos"), (sphx.F_VEL, "crc32_vel"), (sphx.F_DENSITY, "crc32_density")):
                assert _crc_in_particle_order(sphx, gs, f) == st[k], "step %d: %s differs from the reference-source run" % (step, k)
            assert list(gs.iters()) == st["iters_div_den"], (step, gs.iters())
    assert gs.iters() == (20, 2)


def test_reference_source_anchor_statistics_wcsph_pbd_on_gpu(sphx):
    """The ENGINE under its strict contract against the statistics the reference SOURCES gave (refsrc_anchors.json, 9 digits):
    this measures the two documented deviations.  D1 (Tait x^7 as an fp64 multiply chain, reference: powf): the pressure term
    is only non-zero above rest density, so WCSPH agrees to the 9 recorded digits until the column lands (step ~140), stays
    within the north star's 1e-5 to step 200 and then separates like any two roundings of a splashing state (1.6e-4 at step
    300: tools/anchor_probe.py).  D3 (PBD's XSPH evaluated Jacobi-style, reference: in place, i.e. a data race on the GPU and
    ascending index order in a serial build): a real algorithmic difference of order c = 0.05 times the velocity spread --
    1e-4 on the velocity maximum from the first states on, the column height agrees to 1e-6 through step 100."""
    import json, os
    V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refsrc_anchors.json")))["variants"]["float_fabs"]

    def stats(s):
        rho = s.get(sphx.F_DENSITY).astype(np.float64); pos = s.get(sphx.F_POS).astype(np.float64); vel = s.get(sphx.F_VEL).astype(np.float64)
        return {"rho_mean": rho.mean(), "rho_min": rho.min(), "rho_max": rho.max(), "mean_y": pos[:, 1].mean(), "vmax": np.sqrt((vel * vel).sum(1)).max()}

    def run(name, solver, bound):
        A = V[name]
        P, f, b = sphx.scene(24)
        P.solver = solver; P.dt = A["dt"]
        s = sphx.System(P, f, b)
        at, worst = 0, {}
        for st in A["states"]:
            while at < st["step"]:
                s.step(); at += 1
            got = stats(s)
            for k in got:
                dev = abs(got[k] - st[k]) / max(abs(st[k]), 1e-30)
                assert dev <= bound(st["step"], k), "%s step %d %s: %.2e (engine %.9g, reference sources %.9g)" % (name, st["step"], k, dev, got[k], st[k])
                worst[k] = max(worst.get(k, 0.0), dev)
        s.close()
        return worst

    w = run("wcsph", sphx.WCSPH, lambda step, k: 1e-8 if step <= 150 else (1e-5 if step <= 200 else 1e-3))
    assert max(w.values()) > 1e-8, "D1 must show once the column has landed"
    p = run("pbd", sphx.PBD, lambda step, k: {"mean_y": 1e-5 if step <= 100 else 1e-4, "rho_mean": 2e-4}.get(k, 1e-2))
    assert p["vmax"] > 1e-6, "D3 is not a rounding difference"


@pytest.mark.parametrize("solver,dt,first,last", [(0, 0.001, 120, 200), (1, 0.002, 50, 100), (2, 0.002, 50, 80)])
def test_trajectory_bit_exact_through_landing(sphx, oracle, solver, dt, first, last):
    """reference scene, default solver settings (adaptive DFSPH, PBD k = 20): every field bit-identical to the oracle at
    steps first..last (every 5th), which bracket the landing (WCSPH ~140 at dt = 0.001; DFSPH / PBD ~60 at dt = 0.002)"""
    def tweak(P):
        P.dt = dt
    gs, os_, _ = make_pair(sphx, oracle, 24, solver, tweak=tweak)
    names = FIELDS_COMMON + (FIELDS_DFSPH if solver == 1 else []) + (FIELDS_PBD if solver == 2 else [])
    its = set()
    for s in range(1, last + 1):
        gs.step(); os_.step()
        if s >= first and (s % 5 == 0 or s == last):
            compare(sphx, oracle, gs, os_, names, "landing solver %d step %d" % (solver, s))
        if solver == 1:
            assert gs.iters() == os_.iters(), s
            its.add(gs.iters())
    den = gs.get(sphx.F_DENSITY)
    assert den.max() > 0.99, "the column must have landed (densities near rho0)"
    if solver == 1:
        assert (20, 2) in its and (1, 2) in its, "adaptive control must have been exercised up to maxIter"
    if solver == 0:
        assert gs.get(sphx.F_PRESSURE).max() > 0, "Tait pressures must be active"


@pytest.mark.parametrize("mode", ["tail_only", "tail_only_flat_barrier", "gated_only", "windows"])
def test_adaptive_loops_tail_and_gated_launches_bit_exact(sphx, oracle, monkeypatch, mode):
    """adaptive DFSPH through the landing (counts 1 -> 20) with every iteration beyond the reference's minimum inside the persistent
    tail launch (SPHX_DFSPH_WINDOW=0), with gated launches only (SPHX_DFSPH_NO_TAIL=1) and with the default adaptive windows:
    every field bit-identical to the oracle, same iteration counts (DFSPHSolver.cu:187-208, :347-361); the counts read after a
    replayed batch are those of its last step"""
    if mode.startswith("tail_only"):          # (r06: the tail's sweeps are separated by the XCD-hierarchical barrier; _flat_barrier: the r04 one)
        monkeypatch.setenv("SPHX_DFSPH_WINDOW", "0")
    if mode == "tail_only_flat_barrier":
        monkeypatch.setenv("SPHX_DFSPH_TAIL_FLAT", "1")
    if mode == "gated_only":
        monkeypatch.setenv("SPHX_DFSPH_NO_TAIL", "1")
    gs, os_, _ = make_pair(sphx, oracle, 24, 1)
    names = FIELDS_COMMON + FIELDS_DFSPH
    its = []
    for s in range(1, 91):
        gs.step(); os_.step()
        assert gs.iters() == os_.iters(), (mode, s)
        its.append(gs.iters())
        if s % 15 == 0:
            compare(sphx, oracle, gs, os_, names, "%s step %d" % (mode, s))
    assert max(i[0] for i in its) == 20 and min(i[0] for i in its) == 1
    P, f, b = sphx.scene(24); P.solver = sphx.DFSPH
    batch = sphx.System(P, f, b)
    batch.step_n(90)
    assert batch.iters() == its[-1], "counts after a replayed batch"
    assert np.array_equal(batch.get(sphx.F_POS).view(np.uint32), gs.get(sphx.F_POS).view(np.uint32))
    batch.close()


def test_loop_tail_that_cannot_be_resident_reports_a_fault_instead_of_hanging(tmp_path):
    """the persistent tail launch spins on a grid barrier: launched with more blocks than the device holds at once (test build of the
    library, SPHX_DFSPH_TAIL_OVERSUBSCRIBE) the barrier must time out, the step must be reported invalid through the C ABI, and the
    solver must carry on with gated launches -- a hang here would take the device with it"""
    import subprocess, sys, textwrap
    hooks = os.path.join(ROOT, "tests", "libsphx_hooks.so")
    assert os.path.exists(hooks), "tests/libsphx_hooks.so is built by __graft_entry__.build() (make -C tests)"
    script = tmp_path / "oversubscribed.py"
    script.write_text(textwrap.dedent("""
        import sys, os
        sys.path.insert(0, os.path.join(%r, "cpp-fluid-particles_amd"))
        import sphx
        P, f, b = sphx.scene(24); P.solver = sphx.DFSPH          # the reference scene: the loops run long once the column lands (~step 60)
        s = sphx.System(P, f, b)
        failed, after = 0, 0
        for k in range(120):
            try:
                s.step()
                if failed: after += 1
            except RuntimeError as e:
                failed += 1; print("step", k, "reported:", e, flush=True)
            if after >= 10: break
        assert failed == 1 and after == 10, (failed, after)
        assert max(s.iters()) > 2, s.iters()                     # gated launches carry on
        print("OK", flush=True)
    """ % ROOT))
    env = dict(os.environ, SPHX_LIB=hooks, SPHX_DFSPH_TAIL_OVERSUBSCRIBE="1", SPHX_DFSPH_WINDOW="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout and "grid barrier timed out" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_scan_tile_that_never_publishes_is_reported_instead_of_hanging(tmp_path):
    """the one-launch scans of the grid pass (csrc/scan_chain.hpp) wait for the tiles in front of them: with one tile that never
    publishes (test build of the library, SPHX_CHAIN_DROP_TILE) the tiles behind it must give up after their bounded spin, the step
    must be reported invalid through the C ABI, and the next steps must run -- a hang here would take the device with it"""
    import subprocess, sys, textwrap
    hooks = os.path.join(ROOT, "tests", "libsphx_hooks.so")
    assert os.path.exists(hooks), "tests/libsphx_hooks.so is built by __graft_entry__.build() (make -C tests)"
    script = tmp_path / "dropped_tile.py"
    script.write_text(textwrap.dedent("""
        import sys, os, time
        sys.path.insert(0, os.path.join(%r, "cpp-fluid-particles_amd"))
        import sphx
        P, f, b = sphx.scene(24); P.solver = sphx.WCSPH          # 36 k cells = 18 tiles of the cell-table scan
        s = sphx.System(P, f, b)
        s.step_n(5)
        os.environ["SPHX_CHAIN_DROP_TILE"] = "3"
        t0 = time.time(); failed = 0
        try:
            s.step()
        except RuntimeError as e:
            failed += 1; print("reported:", e, flush=True)
        print("the faulty step took %%.1f s" %% (time.time() - t0), flush=True)
        del os.environ["SPHX_CHAIN_DROP_TILE"]
        assert failed == 1
        s.step_n(5); s.step()                                      # the chain re-arms itself: later steps run and report nothing
        print("OK", flush=True)
    """ % ROOT))
    env = dict(os.environ, SPHX_LIB=hooks)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout and "gave up waiting" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_adaptive_row_capacity_grows_and_stays_exact(sphx, oracle):
    """rows start at 48 entries per particle; a state denser than the lattice (two interleaved jittered lattices, ~60 neighbours)
    overflows them: the overflowing particles walk the cells directly (same bits), the builder reports the longest row, the
    host enlarges the rows between steps -- the trajectory equals the oracle bit for bit before, while and after that"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.DFSPH; P.dt = 0.0005
    rng = np.random.default_rng(3)
    a = fluid + rng.uniform(-0.002, 0.002, fluid.shape).astype(np.float32)
    b = fluid[: len(fluid) // 2] + np.float32(0.01) + rng.uniform(-0.002, 0.002, (len(fluid) // 2, 3)).astype(np.float32)
    pos = np.concatenate([a, b]).astype(np.float32)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    longest = []
    for s in range(20):
        gs.step(); os_.step()
        compare(sphx, oracle, gs, os_, ["POS", "VEL", "DENSITY", "KAPPA"], "dense state step %d" % (s + 1))
        assert gs.iters() == os_.iters()
        longest.append(gs.row_stats()[1])
    assert max(longest) > 48, "the state must overflow the initial rows (longest row %d)" % max(longest)
    grown = sphx.row_capacity(gs)
    assert grown >= max(longest[:8]) and grown > 48, "the rows must have been enlarged (capacity %d)" % grown


def test_snapshot_keeps_the_active_count_and_the_pbd_first_step_state(sphx, oracle, tmp_path):
    """snapshot container version 2: (a) a system whose active count was lowered with sphx_set_count continues
    with that count, not with the capacity; (b) a PBD system saved BEFORE its first step still spends that step recording
    positions (PBDSolver.cu:45-49) -- both equal the oracle's uninterrupted run bit for bit"""
    # (a)
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.WCSPH; P.dt = 0.001
    pos, vel = _splash_state(len(fluid), P, 201)
    Po = same_params(oracle.Params(), P)
    a = sphx.System(P, pos, boundary, ctor_step=False)
    o = oracle.System(Po, pos, boundary, ctor_step=False)
    m = len(fluid) - 777
    a.set_count(m); o.set_count(m)
    for _ in range(2):
        a.step(); o.step()
    path = str(tmp_path / "count.bin")
    sphx.save_snapshot(a, path); a.close()
    b = sphx.load_snapshot(path)
    for s_ in range(2):
        b.step(); o.step()
        for nm in ("POS", "VEL", "DENSITY", "ID"):
            assert_bit_equal(b.get(getattr(sphx, "F_" + nm))[:m], o.get(getattr(oracle, "F_" + nm))[:m], "lowered count, resumed step %d %s" % (s_ + 1, nm))
    b.close(); o.close()
    # (b)
    P.solver = sphx.PBD; P.pbd_iters = 3
    Po = same_params(oracle.Params(), P)
    a = sphx.System(P, pos, boundary, ctor_step=False)
    o = oracle.System(Po, pos, boundary, ctor_step=False)
    path = str(tmp_path / "pbd0.bin")
    sphx.save_snapshot(a, path); a.close()
    b = sphx.load_snapshot(path)
    for s_ in range(3):
        b.step(); o.step()
        compare(sphx, oracle, b, o, ["POS", "VEL", "DENSITY", "POS_LAST"], "PBD saved before its first step, step %d" % (s_ + 1))


def test_row_capacity_growth_between_graph_replays(sphx, oracle):
    """the rows are enlarged between two step_n batches, i.e. between hipGraph replays: the reallocation happens in the tuning
    hook (a capture must not allocate) and the next batch re-captures -- results stay oracle-identical (found by bench.py's
    post-impact leg in r03: the first version reallocated lazily, inside the next capture)"""
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.DFSPH; P.dt = 0.0005; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 2
    rng = np.random.default_rng(5)
    a = fluid + rng.uniform(-0.002, 0.002, fluid.shape).astype(np.float32)
    b = fluid[: len(fluid) // 2] + np.float32(0.01) + rng.uniform(-0.002, 0.002, (len(fluid) // 2, 3)).astype(np.float32)
    pos = np.concatenate([a, b]).astype(np.float32)
    Po = same_params(oracle.Params(), P)
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    os_ = oracle.System(Po, pos, boundary, ctor_step=False)
    for batch in range(3):
        gs.step_n(10)
        for _ in range(10):
            os_.step()
        compare(sphx, oracle, gs, os_, ["POS", "VEL", "DENSITY"], "batch %d" % batch)
    assert sphx.row_capacity(gs) > 48


def test_reference_source_anchor_disordered_dfsph_on_gpu(sphx):
    """the ENGINE against the reference-source CRCs of a disordered splash (tests/golden/refsrc_anchors.json, splash_nx12):
    adaptive DFSPH, ragged cells and wall contact from the first step, iteration counts (9,7) .. (4,2)"""
    import json, os
    V = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refsrc_anchors.json")))["variants"]["float_fabs"]["splash_nx12"]["dfsph"]
    P, fluid, boundary = sphx.scene(12)
    P.solver = sphx.DFSPH; P.dt = V["dt"]
    pos, vel = _splash_state(len(fluid), P, V["seed"])
    gs = sphx.System(P, pos, boundary, ctor_step=False)
    gs.set(sphx.F_VEL, vel[gs.get(sphx.F_ID)])
    gs.step()
    states = {st["step"]: st for st in V["states"]}
    for step in range(0, 31):
        if step:
            gs.step()
        st = states.get(step)
        if st:
            for f, k in ((sphx.F_POS, "crc32_pos"), (sphx.F_VEL, "crc32_vel"), (sphx.F_DENSITY, "crc32_density")):
                assert _crc_in_particle_order(sphx, gs, f) == st[k], "step %d: %s differs from the reference-source run" % (step, k)
            assert list(gs.iters()) == st["iters_div_den"], (step, gs.iters())


def test_reference_source_anchor_obstacles_dfsph_on_gpu(sphx, oracle):
    """the ENGINE against the reference-source CRCs of the obstacle scene (refsrc_anchors.json, obstacles_nx12): boundary masses of a
    non-shell boundary set, then adaptive DFSPH driven into the obstacles, iterations (1,2) -> (7,2) -> (20,2)"""
    import json, os, zlib
    from test_cpu_oracle import _obstacle_scene
    A = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refsrc_anchors.json")))["variants"]["float_fabs"]["obstacles_nx12"]
    P, fluid, boundary, vel = _obstacle_scene(sphx, sphx)
    P.solver = sphx.DFSPH; P.dt = A["dfsph"]["dt"]
    gs = sphx.System(P, fluid, boundary, ctor_step=False)
    assert zlib.crc32(gs.get(sphx.F_BMASS).tobytes()) == A["crc32_boundary_mass_sorted"], "boundary masses differ from the reference-source run"
    gs.set(sphx.F_VEL, vel[gs.get(sphx.F_ID)])
    gs.step()
    states = {st["step"]: st for st in A["dfsph"]["states"]}
    for step in range(0, 21):
        if step:
            gs.step()
        st = states.get(step)
        if st:
            for f, k in ((sphx.F_POS, "crc32_pos"), (sphx.F_VEL, "crc32_vel"), (sphx.F_DENSITY, "crc32_density")):
                assert _crc_in_particle_order(sphx, gs, f) == st[k], "step %d: %s differs from the reference-source run" % (step, k)
            assert list(gs.iters()) == st["iters_div_den"], (step, gs.iters())


def test_randomised_parity_stress(sphx, oracle):
    """120 random small cases (tools/stress_parity.py: random container size, solver, constants, state generator -- splash, jittered
    lattice, dense blob, sheets on the walls, particles outside the grid / coincident / on cell faces -- engine schedule, quad / duo
    masks, row capacity): every field bit-identical to the oracle after every step.  (4000 further seeds ran clean in r03:
    profiles/r03_stress_parity.txt.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    keep = {k: os.environ.get(k) for k in ("SPHX_QUAD_MASK", "SPHX_DUO_MASK", "SPHX_NBR_CAP")}
    try:
        failures = [f for f in (mod.run_case(seed) for seed in range(7000, 7120)) if f]
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert not failures, "\n".join(failures)
