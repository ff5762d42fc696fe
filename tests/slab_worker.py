"""worker processes of the slab-decomposition tests (spawned with torch.multiprocessing)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def configure(P, E, solver, adaptive):
    P.solver = {"dfsph": E.DFSPH, "wcsph": E.WCSPH, "pbd": E.PBD}[solver]
    P.pbd_iters = 3
    if solver == "dfsph" and not adaptive:
        P.dfsph_fixed_div, P.dfsph_fixed_den = 2, 3
    P.dt = 0.001


def splash(n, P, seed):
    rng = np.random.default_rng(seed)
    lo = 0.03 * P.space[0]
    pos = rng.uniform(lo, 0.9 * P.space[0], (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(lo, 0.3 * P.space[1], n).astype(np.float32)
    vel = rng.normal(0, 0.6, (n, 3)).astype(np.float32)
    vel[:, 0] += np.where(pos[:, 0] < 0.5 * P.space[0], 2.5, -2.5).astype(np.float32)   # drive flow across the cuts
    return pos, vel


def make_fast(pos, vel, P, every, columns):
    """every `every`-th particle flies `columns` cell columns per step along x, towards the middle of the box (r06: flights past
    the reach of one slab exchange).  In place; the same fp32 arithmetic in every process"""
    fast = np.arange(len(pos)) % every == 0
    vel[fast, 0] = np.where(pos[fast, 0] < 0.5 * P.space[0], 1.0, -1.0).astype(np.float32) * np.float32(columns * P.cell_length / P.dt)
    return vel


def pbd_last_positions(pos, vel, P):
    """PBD derives velocities from displacements: give the particles the splash velocities by moving
    the recorded last positions back by dt * vel (same fp32 arithmetic in every process)"""
    return (pos - np.float32(P.dt) * vel).astype(np.float32)


def run(rank, world, port, backend, engine_kind, nx, steps, outdir, seed, solver="dfsph", adaptive=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import slab_protocol as M
    from slab_cpu_engine import OracleSlabEngine
    if engine_kind == "oracle":
        from oracle import oracle as E
        E.lib().oracle_set_threads(min(E.lib().oracle_max_threads(), 4))     # several ranks share the host
        P, fluid, boundary = E.scene(nx)
    else:
        import sphx as E
        import tuning_env; tuning_env.install(E)
        E.set_device(0)
        torch.cuda.set_device(0)
        E.use_stream(torch.cuda.current_stream().cuda_stream)
        P, fluid, boundary = E.scene(nx)
    configure(P, E, solver, adaptive)
    pos, vel = splash(len(fluid), P, seed)
    bsys = E.System(P, np.zeros((0, 3), np.float32), boundary, ctor_step=False)
    bpos, bmass = bsys.get(E.F_BPOS), bsys.get(E.F_BMASS)
    bsys.close()
    if engine_kind == "oracle":
        make = lambda Pl, cap, bp, bm: OracleSlabEngine(E, Pl, cap, bp, bm)
    else:
        dev = torch.device("cuda", 0)
        make = lambda Pl, cap, bp, bm: M.HipSlabEngine(E, Pl, cap, bp, bm, dev)
    drv, cuts, counts = M.build_slab(make, P, pos, bpos, bmass, rank, world, capacity_factor=2.0, velocity=vel)
    migrated = 0
    prev = None
    for k in range(steps):
        drv.step()
        if solver == "pbd" and k == 0:      # after the recording step (PBDSolver.cu:45-49)
            o0, o1 = drv.owned
            own = drv.e.read("ids", o0, o1).cpu().numpy()
            drv.e.write("pos_last", o0, drv.e.to_device(np.ascontiguousarray(pbd_last_positions(pos, vel, P)[own])))
        ids = drv.owned_state()[0]
        if prev is not None:
            migrated += len(np.setdiff1d(ids, prev))
        prev = ids
    ids, p, v, d = drv.owned_state()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ids=ids, pos=p, vel=v, density=d, cuts=np.array(cuts),
             migrated=np.array(migrated), iters=np.array(drv.iters))
    dist.barrier()
    dist.destroy_process_group()
