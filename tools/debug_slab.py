"""debug helper: run the slab driver with world=2 for the oracle engine and the HIP engine, dump
owned state after every step, report the first (step, field) that differs."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cpp-fluid-particles_amd"))
import torch.multiprocessing as mp
import socket

def port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

def worker(rank, world, prt, kind, nx, steps, outdir, seed):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(prt)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import slab_protocol as M
    from slab_cpu_engine import OracleSlabEngine
    import slab_worker
    if kind == "oracle":
        from oracle import oracle as E
    else:
        import sphx as E
        import tuning_env; tuning_env.install(E)
        E.set_device(0); torch.cuda.set_device(0); E.use_stream(torch.cuda.current_stream().cuda_stream)
    P, fluid, boundary = E.scene(nx)
    P.solver = E.DFSPH; P.dfsph_fixed_div, P.dfsph_fixed_den = 2, 3; P.dt = 0.001
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    bsys = E.System(P, np.zeros((0, 3), np.float32), boundary, ctor_step=False)
    bpos, bmass = bsys.get(E.F_BPOS), bsys.get(E.F_BMASS); bsys.close()
    if kind == "oracle":
        make = lambda Pl, cap, bp, bm: OracleSlabEngine(E, Pl, cap, bp, bm)
    else:
        make = lambda Pl, cap, bp, bm: M.HipSlabEngine(E, Pl, cap, bp, bm, torch.device("cuda", 0))
    drv, cuts, counts = M.build_slab(make, P, pos, bpos, bmass, rank, world, capacity_factor=2.0, velocity=vel)
    # instrument: dump all local arrays (incl ghosts) after chosen phases of step 1
    e = drv.e
    orig_run = e.run
    log = []
    def run(ph):
        orig_run(ph)
        if len(log) < 40:
            n = sum(1 for _ in [0])  # noqa
            c = drv.layers[4] if drv.layers and ph != M.PH_SEARCH else None
            if ph == M.PH_SEARCH:
                c = e.cell_starts([drv.gxl * drv.L])[0]
            if ph == M.PH_SEARCH:
                lay = e.cell_starts([drv.L, (drv.gxl - 1) * drv.L])
            else:
                lay = [drv.layers[0], drv.layers[3]]
            d = {k: e.read(k, lay[0], lay[1]).cpu().numpy().copy() for k in ("ids", "pos", "vel_nbr", "kappa", "warm", "density")}
            log.append((ph, d))
    e.run = run
    drv.step()
    e.run = orig_run
    np.save(os.path.join(outdir, "%s_rank%d.npy" % (kind, rank)), np.array(log, dtype=object), allow_pickle=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    out = tempfile.mkdtemp()
    for kind in ("oracle", "hip"):
        mp.spawn(worker, args=(2, port(), kind, 12, 1, out, 17), nprocs=2, join=True)
    names = ["SEARCH","HEAD","DIV_CORRECT","DIV_ERROR","FORCE","VISC_COLOR","SURFACE","WARM_CORRECT","DEN_ERROR_SET","DEN_CORRECT","DEN_ERROR_ACC","ADVECT"]
    for rank in range(2):
        a = np.load(os.path.join(out, "oracle_rank%d.npy" % rank), allow_pickle=True)
        b = np.load(os.path.join(out, "hip_rank%d.npy" % rank), allow_pickle=True)
        print("rank", rank, "phases logged", len(a), len(b))
        for k, ((pa, da), (pb, db)) in enumerate(zip(a, b)):
            msg = []
            for f in da:
                x, y = da[f], db[f]
                if f == "vel_nbr":
                    y = y[:, :3]
                if x.shape != y.shape:
                    msg.append("%s shape %s vs %s" % (f, x.shape, y.shape)); continue
                xb = x.view(np.uint32) if x.dtype == np.float32 else x
                yb = np.ascontiguousarray(y).view(np.uint32) if y.dtype == np.float32 else y
                bad = np.flatnonzero(xb.reshape(-1) != yb.reshape(-1))
                if bad.size:
                    idx = bad[0] // (x.shape[1] if x.ndim > 1 else 1)
                    msg.append("%s: %d differ, first row %d of %d (id %d) oracle %s hip %s" % (f, bad.size, idx, len(x), da["ids"][idx], x[idx], y[idx]))
            print("  step1 phase#%d %s:" % (k, names[pa]), "OK" if not msg else "; ".join(msg))
