"""One rank of the native slab layer on its RCCL transport (spawned by tests/test_gpu_slab.py, one process per rank).
argv: rank world nx steps seed solver adaptive(0|1) rebalance(0|1) outdir.  The communicator token travels through a
file in outdir, as a launcher's side channel would carry it.  SPHX_TEST_SLABS_PER_PROCESS = L: every process drives L
consecutive slabs (world * L slabs in all), messages between slabs of one process are RCCL sends to self."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: the engine and RCCL must bind to the HIP runtime torch loads, as in bench.py)

import slab_worker  # puts the package on sys.path
import sphx
import tuning_env; tuning_env.install(sphx)      # SPHX_* environment variables -> sphx_tuning (the library reads none itself)


def main():
    rank, world, nx, steps, seed = (int(a) for a in sys.argv[1:6])
    if os.environ.get("SPHX_TEST_WATCHDOG_S"):           # diagnostics: where is this rank after that many seconds?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SPHX_TEST_WATCHDOG_S"]), exit=True)
    solver, adaptive, rebalance, outdir = sys.argv[6], sys.argv[7] == "1", sys.argv[8] == "1", sys.argv[9]
    sphx.set_device(0)
    P, fluid, boundary = sphx.scene(nx)
    slab_worker.configure(P, sphx, solver, adaptive)
    P.reserved[3] = int(os.environ.get("SPHX_TEST_ARITH", "0"))          # arithmetic contract of the slabs (0 strict, 1 tolerance, 2 persistent)
    pos, vel = slab_worker.splash(len(fluid), P, seed)
    if os.environ.get("SPHX_TEST_FAST_EVERY"):           # flights across whole slabs (hop-by-hop exchange)
        slab_worker.make_fast(pos, vel, P, int(os.environ["SPHX_TEST_FAST_EVERY"]), float(os.environ.get("SPHX_TEST_FAST_COLUMNS", "7.2")))
    token_file = os.path.join(outdir, "token")
    if rank == 0:
        token = sphx.rccl_unique_id()
        with open(token_file + ".part", "wb") as f:
            f.write(token)
        os.rename(token_file + ".part", token_file)
    else:
        t0 = time.time()
        while not os.path.exists(token_file):
            if time.time() - t0 > 300:
                raise SystemExit("rank %d: no communicator token" % rank)
            time.sleep(0.05)
        token = open(token_file, "rb").read()
    per = int(os.environ.get("SPHX_TEST_SLABS_PER_PROCESS", "1"))
    g = sphx.SlabGroup(P, pos, boundary, world * per, first_rank=rank * per, local_ranks=per, rccl_id=token, velocity=vel)
    if rebalance:
        g.set_rebalance(1, 0.0)
    cuts = set()
    try:
        for _ in range(steps):
            g.step()
            cuts.update(g.info(k)[:2] for k in range(per))
    except sphx.SphxError as e:       # (the failure tests expect every rank to arrive here together)
        print("rank %d: %s" % (rank, e), flush=True)
        try:
            g.step()
        except sphx.SphxError as e2:
            assert "earlier step" in str(e2), e2
        raise SystemExit(3)
    ids, p, v, d = g.gather_all()
    loaded = sorted({line.split()[-1] for line in open("/proc/self/maps") if "rccl" in line.lower()})      # which RCCL served the calls
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ids=ids, pos=p, vel=v, density=d, iters=np.array(g.iters()),
             distinct_cuts=len(cuts) - per + 1, held=g.info(0)[3], rccl_library=np.array(";".join(loaded)))
    g.close()


if __name__ == "__main__":
    main()
