"""GPU tests of the x-slab decomposition with the HIP engine; the result must equal the single-domain ORACLE
result bit for bit.  Two host drivers: the native layer csrc/slab.hip (loopback transport: all slabs on the test
box's one GPU; its RCCL transport needs a multi-GPU node) and the Python protocol driver tests/slab_protocol.py (ranks are
processes sharing the GPU, talking over gloo)."""
import numpy as np
import pytest

from conftest import assert_bit_equal
import slab_worker
from test_slab_cpu import _free_port, _single_domain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,solver,adaptive", [(2, "dfsph", False), (2, "wcsph", False), (2, "dfsph", True), (2, "pbd", False)])
def test_hip_slab_driver_matches_single_domain_oracle(oracle, tmp_path, world, solver, adaptive):
    import torch.multiprocessing as mp
    nx, steps, seed = 12, 6, 17
    mp.spawn(slab_worker.run, args=(world, _free_port(), "gloo", "hip", nx, steps, str(tmp_path), seed, solver, adaptive),
             nprocs=world, join=True)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    ids = np.concatenate([p["ids"] for p in parts])
    n = len(ids)
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.int32))
    order = np.argsort(ids)
    rp, rv, rd, it = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    if adaptive:
        assert all(tuple(p["iters"]) == it for p in parts)
    assert_bit_equal(np.concatenate([p["pos"] for p in parts])[order], rp, "slab pos")
    assert_bit_equal(np.concatenate([p["vel"] for p in parts])[order], rv, "slab vel")
    assert_bit_equal(np.concatenate([p["density"] for p in parts])[order], rd, "slab density")
    if world > 1:
        assert sum(int(p["migrated"]) for p in parts) > 0


# ------------------------------------------------------------------------------------ the native layer (csrc/slab.hip)
def _native(sphx, world, solver, adaptive, nx, steps, seed, flags, tweak=None, arith=0):
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    P.reserved[3] = arith
    if tweak:
        tweak(P)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    g = sphx.SlabGroup(P, pos, boundary, world, flags=flags, velocity=vel)       # loopback: every slab on this device
    moved = 0
    prev = None
    for _ in range(steps):
        g.step()
        owners = np.concatenate([np.full(g.info(i)[2], i) for i in range(world)])
        ids = np.concatenate([g.gather(i)[0] for i in range(world)])
        own_of = np.empty(len(ids), np.int64); own_of[ids] = owners
        if prev is not None:
            moved += int(np.count_nonzero(own_of != prev))
        prev = own_of
    out = g.gather_all() + (g.iters(), moved)
    g.close()
    return out


@pytest.mark.parametrize("world,solver,adaptive,nx", [(1, "dfsph", False, 12), (2, "dfsph", False, 12), (3, "dfsph", False, 12),
                                                       (8, "dfsph", False, 24), (2, "wcsph", False, 12), (8, "wcsph", False, 24),
                                                       (3, "dfsph", True, 12), (2, "pbd", False, 12), (8, "pbd", False, 24)])
@pytest.mark.parametrize("flags", [0, 1], ids=["overlap", "no-overlap"])
def test_native_slab_layer_matches_single_domain_oracle(sphx, oracle, world, solver, adaptive, nx, flags):
    """csrc/slab.hip with the loopback transport (all slabs on the test box's one GPU): edge-first stages with the
    halo exchange started before the interior sweep (flags 0) and the simple stage-then-exchange schedule (flags 1)
    both reproduce the single-domain ORACLE bit for bit, for 1, 2, 3 and 8 slabs, incl. migration across the cuts,
    adaptive DFSPH iteration counts and PBD's two-column halos"""
    steps, seed = 6, 17
    ids, pos, vel, den, iters, moved = _native(sphx, world, solver, adaptive, nx, steps, seed, flags)
    n = len(ids)
    assert np.array_equal(ids, np.arange(n, dtype=np.int32)), "every particle owned exactly once"
    rp, rv, rd, it = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    assert_bit_equal(pos, rp, "native slab pos"); assert_bit_equal(vel, rv, "native slab vel")
    assert_bit_equal(den, rd, "native slab density")
    if solver == "dfsph":
        assert iters == it
    if world > 1:
        assert moved > 0, "the test must exercise migration across cuts"


def _dense_splash(n, P, seed):
    """the slab tests' splash squeezed to half its height: rows of up to ~70 entries in the first steps (WCSPH relaxes it gently:
    |v| stays below 12, far from the one-column-per-step limit of the exchange)"""
    pos, vel = slab_worker.splash(n, P, seed)
    lo = np.float32(0.03 * P.space[0])
    pos[:, 1] = (lo + (pos[:, 1] - lo) * np.float32(0.5)).astype(np.float32)
    return pos, vel


@pytest.mark.parametrize("world", [1, 2, 3])
def test_slab_rows_grow_like_a_whole_domain_systems(sphx, oracle, world):
    """r05: a slab's neighbour rows start at 48 entries and grow behind the last stage of a step (SPHSystem::phase ->
    BasicSPHSolver::tune; until r05 they were fixed at 96).  A splash whose rows pass 48 entries, long enough for two capacity
    checks (every 8 steps): before the growth the overflowing particles walk the cells, afterwards their rows hold them -- the
    strict result equals the single-domain ORACLE bit for bit throughout, and the rows grew as the whole-domain engine's did
    (the slab that owns the particle with the longest row asks for the same capacity)."""
    nx, steps, seed = 12, 18, 17
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, "wcsph", False)
    pos, vel = _dense_splash(len(fluid), P, seed)
    g = sphx.SlabGroup(P, pos, boundary, world, velocity=vel)
    assert all(g.row_capacity(i) == 48 for i in range(world))
    g.step(steps)
    caps = [g.row_capacity(i) for i in range(world)]
    ids, gp, gv, gd = g.gather_all()
    g.close()
    s = sphx.System(P, pos, boundary, ctor_step=False)
    s.set(sphx.F_VEL, vel[s.get(sphx.F_ID)])
    for _ in range(steps):
        s.step()
    single_cap = sphx.row_capacity(s)
    sp, sv = (s.get(f)[np.argsort(s.get(sphx.F_ID))] for f in (sphx.F_POS, sphx.F_VEL))
    s.close()
    assert single_cap > 48, "the scene must make the whole-domain engine grow its rows"
    assert max(caps) == single_cap and all(48 <= c <= single_cap and c % 4 == 0 for c in caps), (caps, single_cap)
    Q, _, _ = oracle.scene(nx)
    slab_worker.configure(Q, oracle, "wcsph", False)
    o = oracle.System(Q, pos, boundary, ctor_step=False)
    o.set(oracle.F_VEL, vel[o.get(oracle.F_ID)])
    for _ in range(steps):
        o.step()
    order = np.argsort(o.get(oracle.F_ID))
    rp, rv, rd = o.get(oracle.F_POS)[order], o.get(oracle.F_VEL)[order], o.get(oracle.F_DENSITY)[order]
    o.close()
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32))
    assert_bit_equal(sp, rp, "whole-domain engine pos"); assert_bit_equal(sv, rv, "whole-domain engine vel")
    assert_bit_equal(gp, rp, "slab pos"); assert_bit_equal(gv, rv, "slab vel"); assert_bit_equal(gd, rd, "slab density")


def _single_engine(sphx, nx, steps, seed, solver, adaptive, arith, tweak=None):
    """the single-device ENGINE in the given arithmetic on the slab tests' splash state, ordered by particle id"""
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    P.reserved[3] = arith
    if tweak:
        tweak(P)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    s = sphx.System(P, pos, boundary, ctor_step=False)
    s.set(sphx.F_VEL, vel[s.get(sphx.F_ID)])
    for k in range(steps):
        s.step()
        if solver == "pbd" and k == 0:
            s.set(sphx.F_POS_LAST, slab_worker.pbd_last_positions(pos, vel, P)[s.get(sphx.F_ID)])
    order = np.argsort(s.get(sphx.F_ID))
    out = (s.get(sphx.F_POS)[order], s.get(sphx.F_VEL)[order], s.get(sphx.F_DENSITY)[order], s.iters())
    s.close()
    return out


@pytest.mark.parametrize("world,solver,adaptive,nx", [(1, "dfsph", False, 12), (2, "dfsph", False, 12), (3, "dfsph", True, 12), (8, "dfsph", False, 24),
                                                       (2, "wcsph", False, 12), (8, "wcsph", False, 24), (2, "pbd", False, 12), (8, "pbd", False, 24)])
@pytest.mark.parametrize("flags", [0, 1], ids=["overlap", "no-overlap"])
def test_native_slab_layer_in_tolerance_arithmetic(sphx, oracle, monkeypatch, world, solver, adaptive, nx, flags):
    """r05: the slab layer under the TOLERANCE contract (sphx_params.reserved[3] = 1 handed to sphx_slab_create), the
    arithmetic of the headline.  A slab's rows are slices of the single-device rows and both sides pick the same kernel variants at
    these sizes, so the run must equal the single-device tolerance ENGINE bit for bit -- positions, velocities, densities, adaptive
    iteration counts -- for 1, 2, 3 and 8 slabs, all three solvers, both schedules; and it must sit within 1e-5 of the strict ORACLE
    on the first steps while differing from it (the tolerance kernels really ran).  PBD: the single-device engine keeps its skin rows
    off here (slabs rebuild their rows per Jacobi iteration, and under the tolerance contract the order of a row decides the bits)."""
    monkeypatch.setenv("SPHX_PBD_SKIN", "0")
    # (and rows of one fixed capacity on both sides -- no growth at different moments: a particle whose row overflows walks the cells with the plain operators,
    # which is the same bits under the strict contract only -- the splash has rows beyond the single engine's initial 48 entries)
    monkeypatch.setenv("SPHX_NBR_CAP", "96")
    steps, seed = 6, 17
    ids, pos, vel, den, iters, moved = _native(sphx, world, solver, adaptive, nx, steps, seed, flags, arith=1)
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    rp, rv, rd, it = _single_engine(sphx, nx, steps, seed, solver, adaptive, 1)
    assert_bit_equal(pos, rp, "tolerance slab pos"); assert_bit_equal(vel, rv, "tolerance slab vel"); assert_bit_equal(den, rd, "tolerance slab density")
    if solver == "dfsph":
        assert iters == it
    if world > 1:
        assert moved > 0, "the test must exercise migration across cuts"
    op, ov, od = _single_domain(oracle, nx, steps, seed, solver, adaptive)
    assert not np.array_equal(den.view(np.uint32), od.view(np.uint32)), "the tolerance kernels must actually run"
    # (six steps of a disordered splash: the deviation from the strict oracle stays small but is not the subject here)
    assert np.abs(pos.astype(np.float64) - op.astype(np.float64)).max() <= 1e-3 * float(sphx.scene(nx)[0].space[0])


@pytest.mark.parametrize("world,solver", [(1, "dfsph"), (2, "dfsph"), (2, "wcsph"), (2, "pbd")])
def test_scheduled_interior_launches_of_wide_slabs(sphx, oracle, monkeypatch, world, solver):
    """an interior range that is most of a big slab keeps the (y-chunk, x) tile schedule: the launch visits every tile of the
    schedule and the tiles outside the range leave at once.  Forced on for a small scene (the schedule itself and the 3 M
    threshold are both meant for 10 M particles): results must not change."""
    monkeypatch.setenv("SPHX_FORCE_TILE_ORDER", "1")
    monkeypatch.setenv("SPHX_RANGE_ORDER_MIN", "1")
    nx, steps, seed = 16, 6, 29
    ids, pos, vel, den, iters, _ = _native(sphx, world, solver, False, nx, steps, seed, 0)
    rp, rv, rd, it = _single_domain(oracle, nx, steps, seed, solver, False, want_iters=True)
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32))
    assert_bit_equal(pos, rp, "scheduled ranges pos"); assert_bit_equal(vel, rv, "scheduled ranges vel")
    assert_bit_equal(den, rd, "scheduled ranges density")


@pytest.mark.parametrize("solver", ["dfsph", "wcsph", "pbd"])
@pytest.mark.parametrize("flags", [0, 1], ids=["overlap", "no-overlap"])
def test_native_slab_layer_without_surface_effects(sphx, oracle, solver, flags):
    """surface tension and air pressure off: the step takes its other stages (a plain add-delta-v and a separate warm-start
    correction instead of the fused surface sweeps; PBD commits the XSPH velocities in a stage of its own) -- also under
    the two-range edge launches"""
    nx, steps, seed, world = 16, 6, 33, 3

    def tweak(P):
        P.surface_tension = 0.0; P.air_pressure = 0.0
    ids, pos, vel, den, iters, _ = _native(sphx, world, solver, False, nx, steps, seed, flags, tweak)
    P, fluid, boundary = oracle.scene(nx)
    slab_worker.configure(P, oracle, solver, False)
    tweak(P)
    p0, v0 = slab_worker.splash(len(fluid), P, seed)
    s = oracle.System(P, p0, boundary, ctor_step=False)
    s.set(oracle.F_VEL, v0[s.get(oracle.F_ID)])
    for k in range(steps):
        s.step()
        if solver == "pbd" and k == 0:
            s.set(oracle.F_POS_LAST, slab_worker.pbd_last_positions(p0, v0, P)[s.get(oracle.F_ID)])
    order = np.argsort(s.get(oracle.F_ID))
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32))
    assert_bit_equal(pos, s.get(oracle.F_POS)[order], "no-surface slab pos")
    assert_bit_equal(vel, s.get(oracle.F_VEL)[order], "no-surface slab vel")
    assert_bit_equal(den, s.get(oracle.F_DENSITY)[order], "no-surface slab density")


@pytest.mark.parametrize("world,adaptive", [(3, True), (8, False)])
def test_dfsph_stage_order_without_the_edge_stream(sphx, oracle, monkeypatch, world, adaptive):
    """DFSPH slabs sweep the edge layers of a stage on a stream of their own beside the interior (the default, which every other test
    runs); SPHX_SLAB_EDGE_STREAM=0 keeps the serial order edges -> halo -> interior on the engine stream.  Same results."""
    monkeypatch.setenv("SPHX_SLAB_EDGE_STREAM", "0")
    nx, steps, seed = (24 if world == 8 else 12), 6, 19
    ids, pos, vel, den, iters, _ = _native(sphx, world, "dfsph", adaptive, nx, steps, seed, 0)
    rp, rv, rd, it = _single_domain(oracle, nx, steps, seed, "dfsph", adaptive, want_iters=True)
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32))
    assert_bit_equal(pos, rp, "serial edges pos"); assert_bit_equal(vel, rv, "serial edges vel"); assert_bit_equal(den, rd, "serial edges density")
    assert iters == it


def test_native_slab_layer_rejects_bad_geometry(sphx):
    P, fluid, boundary = sphx.scene(8)                 # 9 cell columns: too narrow for 8 slabs
    P.solver = sphx.DFSPH
    with pytest.raises(sphx.SphxError):
        sphx.SlabGroup(P, fluid, boundary, 8)
    with pytest.raises(sphx.SphxError):
        sphx.SlabGroup(P, fluid, boundary, 2, first_rank=1, local_ranks=1)      # a single remote-less slab needs an RCCL token


def test_native_slab_layer_rccl_transport_single_rank(oracle, tmp_path):
    """the RCCL transport with a one-rank communicator (the test box has one GPU and RCCL refuses two ranks on one
    device): library loading, ncclGetUniqueId / ncclCommInitRank, the communication stream and events, and
    ncclAllReduce of the adaptive termination sum all run; the result still equals the oracle.  Runs in a fresh
    process that imports torch first, as bench.py does: RCCL must bind to the HIP runtime the engine uses."""
    import os, subprocess, sys
    nx, steps, seed = 12, 5, 23
    out = str(tmp_path / "rccl1.npz")
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import sphx, slab_worker\n"
        "P, fluid, boundary = sphx.scene(%d)\n"
        "slab_worker.configure(P, sphx, 'dfsph', True)\n"
        "pos, vel = slab_worker.splash(len(fluid), P, %d)\n"
        "g = sphx.SlabGroup(P, pos, boundary, 1, first_rank=0, local_ranks=1, rccl_id=sphx.rccl_unique_id(), velocity=vel)\n"
        "g.step(%d)\n"
        "ids, p, v, d = g.gather_all()\n"
        "np.savez(%r, pos=p, density=d, iters=np.array(g.iters()))\n"
    ) % (os.path.join(slab_worker.ROOT, "cpp-fluid-particles_amd"), os.path.join(slab_worker.ROOT, "tests"), slab_worker.ROOT, nx, seed, steps, out)
    subprocess.check_call([sys.executable, "-c", code])
    z = np.load(out)
    rp, rv, rd, rit = _single_domain(oracle, nx, steps, seed, "dfsph", True, want_iters=True)
    assert_bit_equal(z["pos"], rp, "rccl(1) pos"); assert_bit_equal(z["density"], rd, "rccl(1) density")
    assert tuple(z["iters"]) == rit




def _run_ranks(tmp_path, world, nx, steps, seed, solver, adaptive, rebalance, library, extra_env=None, expect_codes=None):
    """one process per slab on the shared test GPU; returns the per-rank result files"""
    import os, subprocess, sys
    env = dict(os.environ)
    if library:
        env["SPHX_RCCL_LIBRARY"] = library
    else:
        env.pop("SPHX_RCCL_LIBRARY", None)        # the installed librccl
    env.update(extra_env or {})
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(slab_worker.ROOT, "tests"), os.path.join(slab_worker.ROOT, "cpp-fluid-particles_amd"),
                                         slab_worker.ROOT, env.get("PYTHONPATH", "")])
    def launch():
        procs = [subprocess.Popen([sys.executable, os.path.join(slab_worker.ROOT, "tests", "slab_rccl_worker.py"), str(r), str(world), str(nx),
                                   str(steps), str(seed), solver, "1" if adaptive else "0", "1" if rebalance else "0", str(tmp_path)], env=env)
                 for r in range(world)]
        try:
            return [p.wait(timeout=240) for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    codes = launch()
    # (r04 / r05 let a case whose rank had died of a SIGNAL run once more: with a highest-priority edge stream the 8-process
    # late-completion case lost ranks or computed other bits in 1 run of 7.  r06 named the cause -- the deferred mode of the stand-in
    # RCCL beside a highest-priority queue of the same process, profiles/r06_slab_edge_stream.txt -- the product has no such stream any
    # more, and the allowance is gone: a rank that dies fails the test.)
    if expect_codes is not None:
        return codes
    assert codes == [0] * world, "rank exit codes %s" % (codes,)
    return [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,solver,adaptive,rebalance", [(2, "dfsph", False, False), (3, "dfsph", True, True), (4, "wcsph", False, True),
                                                             (3, "pbd", False, True), (8, "dfsph", True, True)])
def test_native_slab_layer_rccl_transport_several_ranks(oracle, tmp_path, world, solver, adaptive, rebalance):
    """the RCCL transport as bench.py --gpus N drives it — one process per slab, every neighbour remote, the token
    handed over a side channel, grouped ncclSend/ncclRecv on the communication stream, ncclAllReduce of the adaptive
    sum, cuts moving while particles migrate — with 2, 3, 4 and 8 ranks.  The box has one GPU and the real RCCL refuses two
    ranks on one device, so the nine RCCL calls are served by tests/mock_rccl.cpp (same matching rules, size
    mismatches and unmatched messages are errors); everything above those calls is the shipped code."""
    import os
    library = os.path.join(slab_worker.ROOT, "tests", "libmock_rccl.so")
    assert os.path.exists(library), "tests/libmock_rccl.so is built by __graft_entry__.build() (make -C tests)"
    nx, steps, seed = (32 if world == 8 else 16), 7, 31          # 8 ranks: the shape of the driver's --gpus 8 run
    parts = _run_ranks(tmp_path, world, nx, steps, seed, solver, adaptive, rebalance, library)
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    order = np.argsort(ids)
    rp, rv, rd, rit = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    assert_bit_equal(np.concatenate([p["pos"] for p in parts])[order], rp, "rccl ranks pos")
    assert_bit_equal(np.concatenate([p["vel"] for p in parts])[order], rv, "rccl ranks vel")
    assert_bit_equal(np.concatenate([p["density"] for p in parts])[order], rd, "rccl ranks density")
    if solver == "dfsph":
        assert all(tuple(p["iters"]) == rit for p in parts)
    if rebalance and world < 8 and solver != "pbd":     # (8 narrow slabs, or PBD's two ghost columns on this 17-column grid: the rule's
        assert any(int(p["distinct_cuts"]) > 1 for p in parts), "the cuts must have moved"   # minimum width keeps the cuts where they are;
                                                                                             # test_native_slab_layer_moving_cuts[pbd] moves them)


def test_bench_launch_line_two_ranks(tmp_path):
    """the driver's N > 1 command line — python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 — end to
    end on the one-GPU box: both ranks on device 0 (SPHX_BENCH_DEVICE), the RCCL calls served by the test stand-in.
    Rank 0 must print exactly one JSON line that carries the contract's keys for the whole job."""
    import json, os, subprocess, sys
    from test_slab_cpu import _free_port
    env = dict(os.environ)
    env.update(SPHX_RCCL_LIBRARY=os.path.join(slab_worker.ROOT, "tests", "libmock_rccl.so"), SPHX_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(slab_worker.ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--nx", "40"]
    out = subprocess.run(cmd, env=env, cwd=slab_worker.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 2 and r["unit"] == "steps/s" and r["scaling"] == "strong"
    assert r["value"] > 0 and abs(r["value"] * r["ms_per_step"] - 1000.0) < 1.0
    assert r["config"]["particles"] == 40 * 60 * 40 and "2 x-slabs" in r["config"]["decomposition"]
    # r05: the slabs run the headline's arithmetic contract (bench.py --arith persistent by default -> tolerance in the slab layer) and the
    # line carries its own base: the same workload as ONE slab in the same arithmetic, measured in the same run
    assert r["config"]["arithmetic"] == "tolerance" and r["config"]["arithmetic_asked"] == "persistent"
    assert r["scaling_base"]["ms_per_step"] > 0 and "ONE slab" in r["scaling_base"]["what"] and "tolerance" in r["scaling_base"]["what"]
    assert abs(r["scaling_base"]["speedup_of_this_line"] * r["ms_per_step"] - r["scaling_base"]["ms_per_step"]) < 1e-6 * r["scaling_base"]["ms_per_step"] + 1e-9
    assert r["roofline"] and r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    # the line verifies itself: the communicator as the RCCL library reports it, every rank's device, clock, ownership,
    # traffic and roofline leg
    assert r["rccl"] == {"ranks": 2, "transport": "rccl", "world_size_launcher": 2, "distinct_devices": 1}      # (one-GPU box: both ranks on device 0)
    assert [q["rank"] for q in r["ranks"]] == [0, 1] and [q["rccl"]["comm_rank"] for q in r["ranks"]] == [0, 1]
    for q in r["ranks"]:
        assert q["rccl"]["comm_ranks"] == 2 and q["rccl"]["transport"] == "rccl"
        assert len(q["device_pci_id"]) >= 7 and q["device_name"]
        assert q["ms_per_step"] > 0 and q["host_wait_seconds"] >= 0
        assert q["halo_bytes_sent_per_step"] > 0 and q["halo_bytes_received_per_step"] > 0 and q["exchanges_per_step"] >= 10
        assert q["roofline"] and 0 < q["roofline"]["frac"] < 1
    assert r["ranks"][0]["halo_bytes_sent_per_step"] == r["ranks"][1]["halo_bytes_received_per_step"]
    assert sum(sl["owned"] for q in r["ranks"] for sl in q["slabs"]) == r["owned_particles"]["total"] == 40 * 60 * 40
    assert r["rank_ms_per_step"]["max"] <= r["ms_per_step"] * 1.001 and r["rank_ms_per_step"]["min"] > 0


@pytest.mark.parametrize("solver,adaptive", [("dfsph", False), ("wcsph", False), ("pbd", False), ("dfsph", True)])
def test_native_slab_layer_moving_cuts(sphx, oracle, solver, adaptive):
    """cut re-balancing forced to act every step with zero tolerance (4 slabs, a splash that sloshes along x): columns
    change owners while particles migrate; the result still equals the single-domain oracle bit for bit"""
    nx, steps, seed, world = 24, 8, 29, 4
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    g = sphx.SlabGroup(P, pos, boundary, world, velocity=vel)
    g.set_rebalance(1, 0.0)
    cuts0 = [g.info(i)[:2] for i in range(world)]
    seen = set()
    for _ in range(steps):
        g.step()
        cuts = tuple(g.info(i)[:2] for i in range(world))
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1)), "neighbours must agree on every cut: %s" % (cuts,)
        assert cuts[0][0] == 0 and cuts[-1][1] == P.cells[0]
        seen.add(cuts)
    ids, p, v, d = g.gather_all()
    it = g.iters()
    g.close()
    assert len(seen) > 1, "the cuts must have moved (initial %s)" % (cuts0,)
    rp, rv, rd, rit = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32))
    assert_bit_equal(p, rp, "moving cuts pos"); assert_bit_equal(v, rv, "moving cuts vel"); assert_bit_equal(d, rd, "moving cuts density")
    if solver == "dfsph":
        assert it == rit


@pytest.mark.parametrize("solver,adaptive", [("dfsph", False), ("wcsph", False), ("dfsph", True)])
def test_native_slab_layer_jumps_of_several_columns(sphx, oracle, solver, adaptive):
    """r06: the exchange reaches as far as the neighbouring slab is wide (minus its far edge columns), not one column.
    Every fifth particle of a splash flies 2.5 cell columns per step along x (|v| dt = 2.5 cell lengths), the others slosh as usual:
    owners change two and three columns behind the cuts, ghosts appear and vanish within one step -- positions, velocities,
    densities and iteration counts still equal the single-domain ORACLE bit for bit (3 slabs, the middle one with neighbours on
    both sides).  /root/reference/src/SPHSystem.cu:114-127 re-bins any displacement; so does the slab layer within that reach."""
    nx, steps, seed, world = 24, 4, 41, 3
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    slab_worker.make_fast(pos, vel, P, 5, 2.5)
    g = sphx.SlabGroup(P, pos, boundary, world, velocity=vel)
    cuts = [g.info(i)[:2] for i in range(world)]
    far = 0
    prev_col = prev_own = None
    for _ in range(steps):
        g.step()
        parts = [g.gather(i) for i in range(world)]
        ids = np.concatenate([q[0] for q in parts]); xs = np.concatenate([q[1][:, 0] for q in parts])
        owners = np.concatenate([np.full(len(q[0]), i) for i, q in enumerate(parts)])
        col = np.empty(len(ids), np.int64); own = np.empty(len(ids), np.int64)
        col[ids] = (xs / np.float32(P.cell_length)).astype(np.int64); own[ids] = owners
        if prev_col is not None:
            far += int(np.count_nonzero((np.abs(col - prev_col) >= 2) & (own != prev_own)))
        prev_col, prev_own = col, own
    ids, p, v, d = g.gather_all()
    it = g.iters()
    g.close()
    assert far > 20, "the test must move particles two and more columns across a cut (cuts %s, %d such moves)" % (cuts, far)
    Po, _, bo = oracle.scene(nx)
    slab_worker.configure(Po, oracle, solver, adaptive)
    o = oracle.System(Po, pos, bo, ctor_step=False)
    oid = o.get(oracle.F_ID)
    o.set(oracle.F_VEL, vel[oid])
    for k in range(steps):
        o.step()
        if solver == "pbd" and k == 0:
            o.set(oracle.F_POS_LAST, slab_worker.pbd_last_positions(pos, vel, Po)[o.get(oracle.F_ID)])
    order = np.argsort(o.get(oracle.F_ID))
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    assert_bit_equal(p, o.get(oracle.F_POS)[order], "far jumps pos"); assert_bit_equal(v, o.get(oracle.F_VEL)[order], "far jumps vel")
    assert_bit_equal(d, o.get(oracle.F_DENSITY)[order], "far jumps density")
    if solver == "dfsph":
        assert it == o.iters()


@pytest.mark.parametrize("solver,adaptive,world", [("dfsph", False, 4), ("wcsph", False, 5), ("dfsph", True, 5), ("wcsph", False, 8)])
def test_native_slab_layer_flights_across_whole_slabs(sphx, oracle, solver, adaptive, world):
    """... and beyond that reach the rows travel hop by hop inside the step (sphx_slab_group::forwardFarFlyers): every 40th particle of the
    splash flies 7.2 columns per step -- across slabs of 5-6 columns, into their far edge columns, two slabs away, against the walls --
    and the run still equals the single-domain ORACLE bit for bit: ownership, ghosts of every slab on the way, the pre-sort order of the
    stable cell sort, adaptive iteration counts.  The under-resolved impact of the 10.3 M scene does exactly this (|v| to 1700 m/s under
    the reference's adaptive control: profiles/r06_vmax_probe.txt); the plain engine never cared, now the slab layer does not either."""
    nx, steps, seed = 24, 4, 43
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    slab_worker.make_fast(pos, vel, P, 40, 7.2)
    g = sphx.SlabGroup(P, pos, boundary, world, velocity=vel)
    cuts = [g.info(i)[:2] for i in range(world)]
    hops = 0
    prev_own = None
    for _ in range(steps):
        g.step()
        parts = [g.gather(i) for i in range(world)]
        ids = np.concatenate([q[0] for q in parts])
        owners = np.concatenate([np.full(len(q[0]), i) for i, q in enumerate(parts)])
        own = np.empty(len(ids), np.int64); own[ids] = owners
        if prev_own is not None:
            hops += int(np.count_nonzero(np.abs(own - prev_own) >= 2))
        prev_own = own
    ids, p, v, d = g.gather_all()
    it = g.iters()
    g.close()
    assert hops > 10, "the test must send particles two and more slabs away in one step (cuts %s, %d such moves)" % (cuts, hops)
    Po, _, bo = oracle.scene(nx)
    slab_worker.configure(Po, oracle, solver, adaptive)
    o = oracle.System(Po, pos, bo, ctor_step=False)
    o.set(oracle.F_VEL, vel[o.get(oracle.F_ID)])
    for k in range(steps):
        o.step()
        if solver == "pbd" and k == 0:
            o.set(oracle.F_POS_LAST, slab_worker.pbd_last_positions(pos, vel, Po)[o.get(oracle.F_ID)])
    order = np.argsort(o.get(oracle.F_ID))
    assert np.array_equal(ids, np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    assert_bit_equal(p, o.get(oracle.F_POS)[order], "far flights pos"); assert_bit_equal(v, o.get(oracle.F_VEL)[order], "far flights vel")
    assert_bit_equal(d, o.get(oracle.F_DENSITY)[order], "far flights density")
    if solver == "dfsph":
        assert it == o.iters()


@pytest.mark.parametrize("world,solver,adaptive,library", [(4, "dfsph", True, "mock-deferred"), (5, "wcsph", False, "mock"), (4, "dfsph", False, "mock-deferred")])
def test_far_flights_over_the_rccl_transport(oracle, tmp_path, world, solver, adaptive, library):
    """the hop-by-hop exchange between PROCESSES (one per slab, the stand-in RCCL, transfers landing 300 us late in the deferred form):
    grouped send/recv per hop, the blocking all-reduce that ends the hops, moving cuts -- oracle-identical"""
    nx, steps, seed = 24, 4, 43
    env = {"SPHX_TEST_FAST_EVERY": "40", "SPHX_TEST_FAST_COLUMNS": "7.2"}
    if library == "mock-deferred":
        env["SPHX_MOCK_RCCL_DEFER_US"] = "300"
    parts = _run_ranks(tmp_path, world, nx, steps, seed, solver, adaptive, True, _mock_library(), env)
    ids = np.concatenate([q["ids"] for q in parts]); order = np.argsort(ids)
    assert np.array_equal(ids[order], np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    Po, fluid, bo = oracle.scene(nx)
    slab_worker.configure(Po, oracle, solver, adaptive)
    pos, vel = slab_worker.splash(len(fluid), Po, seed)
    slab_worker.make_fast(pos, vel, Po, 40, 7.2)
    o = oracle.System(Po, pos, bo, ctor_step=False)
    o.set(oracle.F_VEL, vel[o.get(oracle.F_ID)])
    for k in range(steps):
        o.step()
        if solver == "pbd" and k == 0:
            o.set(oracle.F_POS_LAST, slab_worker.pbd_last_positions(pos, vel, Po)[o.get(oracle.F_ID)])
    oo = np.argsort(o.get(oracle.F_ID))
    for name, f in (("pos", oracle.F_POS), ("vel", oracle.F_VEL), ("density", oracle.F_DENSITY)):
        assert_bit_equal(np.concatenate([q[name] for q in parts])[order], o.get(f)[oo], "far flights over the transport: " + name)
    if solver == "dfsph":
        assert all(tuple(q["iters"]) == o.iters() for q in parts)


def test_pbd_slabs_still_refuse_travel_beyond_their_ghost_columns(sphx):
    """PBD sweeps run on positions that moved inside the step over the cell table of the step's start (PBDSolver.cu:139-141): a slab's
    two ghost columns cover one column of travel.  A particle that moved farther is found by the next exchange and ends the run on
    every slab together -- hop-by-hop delivery would not repair the sweeps that already ran"""
    nx, world = 24, 3
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, "pbd", False)
    pos, vel = slab_worker.splash(len(fluid), P, 7)
    slab_worker.make_fast(pos, vel, P, 50, 2.5)
    g = sphx.SlabGroup(P, pos, boundary, world, velocity=vel)
    with pytest.raises(sphx.SphxError, match="PBD: a particle moved more than one cell column"):
        for _ in range(4):
            g.step()
    g.close()


def test_native_slab_layer_config3_size_matches_single_system(sphx):
    """BASELINE config 3's size (1,022,208 particles, DFSPH v=1 d=4) over 8 loopback slabs with cuts re-balanced every
    step: bit-identical to the plain single-device system (itself oracle-identical at small sizes)"""
    P, fluid, boundary = sphx.scene(88)
    P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
    steps = 4
    ref = sphx.System(P, fluid, boundary)            # constructor step = step 1
    ref.step_n(steps - 1)
    order = np.argsort(ref.get(sphx.F_ID))
    rp, rv, rd = ref.get(sphx.F_POS)[order], ref.get(sphx.F_VEL)[order], ref.get(sphx.F_DENSITY)[order]
    ref.close()
    g = sphx.SlabGroup(P, fluid, boundary, 8)
    g.set_rebalance(1, 0.0)
    g.step(steps)
    ids, p, v, d = g.gather_all()
    held = sum(g.info(i)[3] for i in range(8)); owned = sum(g.info(i)[2] for i in range(8))
    g.close()
    assert owned == len(fluid) and held > owned
    assert np.array_equal(ids, np.arange(len(fluid), dtype=np.int32))
    assert_bit_equal(p, rp, "8 slabs pos"); assert_bit_equal(v, rv, "8 slabs vel"); assert_bit_equal(d, rd, "8 slabs density")


def test_native_slab_layer_config5_form_matches_single_system(sphx):
    """BASELINE config 5 in its own form: 10,288,500 particles (nx = 190), DFSPH v=1 d=4, EIGHT x-slabs (loopback: the box
    has one GPU) with the cuts re-balanced every step -- bit-identical to the plain single-device 10 M system after 3 steps.
    Partition rule = the engine's cell-column expression (CUDAFunctions.cuh:64-70)."""
    P, fluid, boundary = sphx.scene(190)
    P.solver = sphx.DFSPH; P.dfsph_fixed_div = 1; P.dfsph_fixed_den = 4
    steps = 3
    ref = sphx.System(P, fluid, boundary)            # constructor step = step 1
    ref.step_n(steps - 1)
    order = np.argsort(ref.get(sphx.F_ID))
    rp, rv, rd = ref.get(sphx.F_POS)[order], ref.get(sphx.F_VEL)[order], ref.get(sphx.F_DENSITY)[order]
    ref.close()
    g = sphx.SlabGroup(P, fluid, boundary, 8)
    g.set_rebalance(1, 0.0)
    cuts = set()
    for _ in range(steps):
        g.step(1)
        cuts.add(tuple(g.info(i)[:2] for i in range(8)))
    ids, p, v, d = g.gather_all()
    owned = [g.info(i)[2] for i in range(8)]; held = [g.info(i)[3] for i in range(8)]
    g.close()
    assert sum(owned) == len(fluid) == 10288500 and all(h > o for h, o in zip(held, owned))
    assert len(cuts) > 1, "the cuts must have moved"
    assert np.array_equal(ids, np.arange(len(fluid), dtype=np.int32))
    assert_bit_equal(p, rp, "config 5, 8 slabs pos"); assert_bit_equal(v, rv, "config 5, 8 slabs vel"); assert_bit_equal(d, rd, "config 5, 8 slabs density")


def test_native_slab_layer_rccl_transport_8_ranks_1m(sphx, tmp_path):
    """the 8-process RCCL-transport run (stand-in library, see above) at BASELINE config 3's size, nx = 88: 1,022,208
    particles of a disordered splash, DFSPH fixed iterations, cuts moving every step; equals the single-device ENGINE
    (itself oracle-identical) bit for bit"""
    import os
    library = os.path.join(slab_worker.ROOT, "tests", "libmock_rccl.so")
    nx, steps, seed, solver = 88, 4, 53, "dfsph"
    parts = _run_ranks(tmp_path, 8, nx, steps, seed, solver, False, True, library)
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    order = np.argsort(ids)
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, False)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    s = sphx.System(P, pos, boundary, ctor_step=False)
    s.set(sphx.F_VEL, vel[s.get(sphx.F_ID)])
    for _ in range(steps):
        s.step()
    o = np.argsort(s.get(sphx.F_ID))
    assert_bit_equal(np.concatenate([p["pos"] for p in parts])[order], s.get(sphx.F_POS)[o], "8 ranks 1M pos")
    assert_bit_equal(np.concatenate([p["vel"] for p in parts])[order], s.get(sphx.F_VEL)[o], "8 ranks 1M vel")
    assert_bit_equal(np.concatenate([p["density"] for p in parts])[order], s.get(sphx.F_DENSITY)[o], "8 ranks 1M density")


# ---- the same transport under REAL asynchrony: the stand-in's deferred mode (ncclGroupEnd only enqueues; data lands late, in
# stream order, behind a spin kernel) -- what the overlap schedule has to survive on xGMI
def _mock_library():
    import os
    library = os.path.join(slab_worker.ROOT, "tests", "libmock_rccl.so")
    assert os.path.exists(library), "tests/libmock_rccl.so is built by __graft_entry__.build() (make -C tests)"
    return library


def _hooks_library():
    """the TEST build of the engine (slab.hip under -DSPHX_TEST_HOOKS): the only library that reads SPHX_SLAB_FAULT"""
    import os
    library = os.path.join(slab_worker.ROOT, "tests", "libsphx_hooks.so")
    assert os.path.exists(library), "tests/libsphx_hooks.so is built by __graft_entry__.build() (make -C tests)"
    return library


def _compare_with_oracle(oracle, parts, nx, steps, seed, solver, adaptive):
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    order = np.argsort(ids)
    rp, rv, rd, rit = _single_domain(oracle, nx, steps, seed, solver, adaptive, want_iters=True)
    same = all(np.array_equal(np.concatenate([p[k] for p in parts])[order].view(np.uint32), r.view(np.uint32))
               for k, r in (("pos", rp), ("vel", rv), ("density", rd)))
    return same, rit


@pytest.mark.parametrize("world,solver,adaptive,rebalance", [(2, "dfsph", False, False), (3, "dfsph", True, True), (4, "wcsph", False, True),
                                                             (3, "pbd", False, True), (8, "dfsph", False, True)])
def test_native_slab_layer_rccl_transport_deferred_completion(oracle, tmp_path, world, solver, adaptive, rebalance):
    """several ranks, transfers completing 300 us AFTER ncclGroupEnd returned, on the communication stream: results still
    equal the single-domain oracle bit for bit (every consumer of a halo is ordered after its transfer, no send buffer is
    overwritten before it was read)"""
    nx, steps, seed = (32 if world == 8 else 16), 6, 41
    parts = _run_ranks(tmp_path, world, nx, steps, seed, solver, adaptive, rebalance, _mock_library(), {"SPHX_MOCK_RCCL_DEFER_US": "300"})
    same, rit = _compare_with_oracle(oracle, parts, nx, steps, seed, solver, adaptive)
    assert same, "deferred completion changed the results"
    if solver == "dfsph":
        assert all(tuple(p["iters"]) == rit for p in parts)


@pytest.mark.parametrize("world,solver,adaptive,library", [(3, "dfsph", True, "mock-deferred"), (4, "wcsph", False, "mock-deferred"), (8, "dfsph", False, "mock-deferred"),
                                                           (4, "dfsph", True, "installed-rccl-to-self")])
def test_native_slab_layer_tolerance_arithmetic_over_the_rccl_transport(sphx, tmp_path, monkeypatch, world, solver, adaptive, library):
    """the tolerance contract over the RCCL transport: 3, 4 and 8 processes with transfers landing 300 us late (stand-in library), and
    ONE process driving 4 slabs through the installed librccl (grouped sends to self) -- each bit-identical to the single-device
    tolerance engine, iteration counts included, with moving cuts"""
    nx, steps, seed = (32 if world == 8 else (24 if library.startswith("installed") else 16)), 6, 41
    monkeypatch.setenv("SPHX_NBR_CAP", "96")          # one fixed row capacity for the slabs and the single-device reference (see above)
    if library.startswith("installed"):
        parts = _run_ranks(tmp_path, 1, nx, steps, seed, solver, adaptive, True, None, {"SPHX_TEST_SLABS_PER_PROCESS": str(world), "SPHX_TEST_ARITH": "1"})
        assert "mock" not in str(parts[0]["rccl_library"]) and "librccl" in str(parts[0]["rccl_library"])
    else:
        parts = _run_ranks(tmp_path, world, nx, steps, seed, solver, adaptive, True, _mock_library(), {"SPHX_MOCK_RCCL_DEFER_US": "300", "SPHX_TEST_ARITH": "1"})
    ids = np.concatenate([p["ids"] for p in parts])
    assert np.array_equal(np.sort(ids), np.arange(len(ids), dtype=np.int32)), "every particle owned exactly once"
    order = np.argsort(ids)
    rp, rv, rd, rit = _single_engine(sphx, nx, steps, seed, solver, adaptive, 1)
    assert_bit_equal(np.concatenate([p["pos"] for p in parts])[order], rp, "tolerance ranks pos")
    assert_bit_equal(np.concatenate([p["vel"] for p in parts])[order], rv, "tolerance ranks vel")
    assert_bit_equal(np.concatenate([p["density"] for p in parts])[order], rd, "tolerance ranks density")
    if solver == "dfsph":
        assert all(tuple(p["iters"]) == rit for p in parts)


@pytest.mark.parametrize("procs,per,solver,adaptive,defer", [(2, 2, "dfsph", True, "0"), (2, 3, "pbd", False, "300"), (4, 2, "wcsph", False, "300")])
def test_native_slab_layer_rccl_transport_several_slabs_per_process(oracle, tmp_path, procs, per, solver, adaptive, defer):
    """a process may drive several consecutive slabs over the RCCL transport: slab r lives in process r / per, messages between
    slabs of one process are sends to self, both sides post the messages of a process pair in (from, to) order.  Stand-in
    library, immediate and deferred completion, cuts moving."""
    nx, steps, seed = 24, 6, 43
    parts = _run_ranks(tmp_path, procs, nx, steps, seed, solver, adaptive, True, _mock_library(),
                       {"SPHX_TEST_SLABS_PER_PROCESS": str(per), "SPHX_MOCK_RCCL_DEFER_US": defer})
    same, rit = _compare_with_oracle(oracle, parts, nx, steps, seed, solver, adaptive)
    assert same, "several slabs per process changed the results"
    if solver == "dfsph":
        assert all(tuple(p["iters"]) == rit for p in parts)


@pytest.mark.parametrize("slabs,solver,adaptive", [(4, "dfsph", True), (3, "pbd", False), (5, "wcsph", False)])
def test_native_slab_layer_on_the_installed_rccl_sends_to_self(oracle, tmp_path, slabs, solver, adaptive):
    """the REAL librccl on the one-GPU box: one process, a one-rank communicator, 3-5 slabs -- so every size message,
    migration payload and halo of the protocol is a real grouped ncclSend / ncclRecv (to self) on the communication stream,
    completing when RCCL completes it, and the engine streams are ordered against it by the layer's own events: the
    edge-first overlap schedule, moving cuts and the adaptive all-reduce included.  Bit-identical to the oracle.
    (What this cannot show is a transfer between two devices; everything on the calling side of RCCL is the shipped code.)"""
    nx, steps, seed = 24, 6, 47
    parts = _run_ranks(tmp_path, 1, nx, steps, seed, solver, adaptive, True, None, {"SPHX_TEST_SLABS_PER_PROCESS": str(slabs)})
    same, rit = _compare_with_oracle(oracle, parts, nx, steps, seed, solver, adaptive)
    library = str(parts[0]["rccl_library"])
    assert "librccl" in library and "mock" not in library, "the installed RCCL must have served the calls: " + library
    assert same, "the run over the installed RCCL differs from the oracle"
    if solver == "dfsph":
        assert tuple(parts[0]["iters"]) == rit
    assert int(parts[0]["distinct_cuts"]) > 1 or solver == "pbd", "the cuts must have moved"


@pytest.mark.parametrize("solver", ["dfsph", "wcsph"])
def test_a_missing_transport_wait_is_detected(oracle, tmp_path, solver):
    """NEGATIVE test: with SPHX_SLAB_FAULT=skipwait the engine stream is not ordered after the overlapped halo transfers.
    Under deferred completion the results must then DIFFER from the oracle -- i.e. the several-ranks tests above would
    catch a wait() that the edge-first schedule forgot.  (With the immediate stand-in the same fault goes unnoticed,
    which is asserted too: that is the blind spot VERDICT r02 named.)"""
    nx, steps, seed = 16, 6, 41
    codes = _run_ranks(tmp_path, 2, nx, steps, seed, solver, False, False, _mock_library(),
                       {"SPHX_MOCK_RCCL_DEFER_US": "2000", "SPHX_SLAB_FAULT": "skipwait", "SPHX_LIB": _hooks_library()}, expect_codes=True)
    if codes == [0, 0]:          # the run survived its stale ghosts: then its results must be wrong
        parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
        same, _ = _compare_with_oracle(oracle, parts, nx, steps, seed, solver, False)
        assert not same, "a dropped wait() went unnoticed under deferred completion"
    else:                        # ... or the garbage tripped the layer's own checks ("farther in one step than the exchange reaches"): detected as well
        assert all(c in (0, 3) for c in codes), codes
    blind = tmp_path / "immediate"; blind.mkdir()
    parts = _run_ranks(blind, 2, nx, steps, seed, solver, False, False, _mock_library(), {"SPHX_SLAB_FAULT": "skipwait", "SPHX_LIB": _hooks_library()})
    same, _ = _compare_with_oracle(oracle, parts, nx, steps, seed, solver, False)
    assert same, "the immediate stand-in completes inside ncclGroupEnd: the fault cannot show there"


def test_rank_local_failure_stops_every_rank(tmp_path):
    """a capacity overflow on ONE rank (injected on rank 1 at step 2) must end the step on EVERY rank with an
    error instead of leaving the neighbours waiting in ncclRecv: all three processes exit with the worker's error code
    well inside the time limit, and none of them hangs."""
    import time
    t0 = time.time()
    codes = _run_ranks(tmp_path, 3, 16, 5, 41, "dfsph", False, False, _mock_library(), {"SPHX_SLAB_FAULT": "capacity:1:2", "SPHX_LIB": _hooks_library()}, expect_codes=True)
    assert time.time() - t0 < 120, "the ranks must not wait for a timeout"
    assert all(c == 3 for c in codes), "every rank reports the failure (exit code 3 = SphxError): %s" % (codes,)


def test_randomised_slab_stress(sphx, oracle):
    """40 random slab decompositions (tools/stress_slab.py: random container, 1-8 loopback slabs, solver, adaptive / fixed DFSPH,
    overlap on / off, re-balancing cadence, splash state) equal the single-domain oracle bit for bit.  (1800 further seeds ran clean
    in r03: profiles/r03_stress_slab.txt.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_slab", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_slab.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    failures = [f for f in (mod.run_case(seed) for seed in range(5000, 5040)) if f]
    assert not failures, "\n".join(failures)
