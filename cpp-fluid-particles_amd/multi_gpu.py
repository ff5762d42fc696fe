"""multi_gpu.py — launcher of the native x-slab layer (csrc/slab.hip) for `bench.py --gpus N`.

One process per GPU.  The data path is C++: RCCL point-to-point halos over xGMI (or the loopback transport), device-side
particle exchange, edge-first stages so that halo traffic overlaps the interior sweeps.  This file only
bootstraps it: the 128-byte RCCL token and the closing barrier / max-over-ranks travel over a gloo side group.
(The torch.distributed restatement of the protocol that used to live here is test infrastructure: tests/slab_protocol.py.)
"""
import os
import time

import torch
import torch.distributed as dist


def _bootstrap(rank, world):
    """a side channel for the 128-byte RCCL token and the closing barrier / max-over-ranks: a gloo group over TCP
    (the data path itself never touches torch.distributed)"""
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world)


def run_slab_bench(args, rank, world, local_rank):
    """bench.py --gpus N (N > 1): the same global workload split into N x-slabs (strong scaling), driven by the
    native layer csrc/slab.hip: one process per GPU, RCCL point-to-point halos over xGMI, edge-first stages so that
    halo traffic overlaps the interior sweeps.  With one process (--force-slab) `args.slabs` loopback slabs share
    the device, which measures the decomposition's overhead without a second GPU."""
    import sphx
    torch.cuda.set_device(local_rank)
    sphx.set_device(local_rank)
    # The edge stream of the slab layer is a default-priority stream (r06: the switch for a highest-priority one is gone
    # from the product -- profiles/r06_slab_edge_stream.txt names what made ranks of the 8-process test fail with it).
    P, fluid, boundary = sphx.scene(args.nx)
    solver_name = getattr(args, "solver", "dfsph")
    P.solver = {"wcsph": sphx.WCSPH, "dfsph": sphx.DFSPH, "pbd": sphx.PBD}[solver_name]
    P.dfsph_fixed_div, P.dfsph_fixed_den, P.pbd_iters = args.div_iters, args.den_iters, getattr(args, "pbd_iters", 4)
    if P.solver == sphx.WCSPH:
        P.dt = 0.001
    # arithmetic contract of the slabs: the headline's (bench.py --arith).  The slab layer runs strict and tolerance arithmetic; rows that
    # persist across steps are a whole-domain mode, so "persistent" runs as tolerance here and the line says so.
    arith_asked = getattr(args, "arith", "strict")
    arith_ran = "tolerance" if arith_asked == "persistent" else arith_asked
    P.reserved[3] = {"strict": 0, "tolerance": 1}[arith_ran]
    flags = sphx.SLAB_NO_OVERLAP if getattr(args, "no_overlap", False) else 0
    if world > 1:
        _bootstrap(rank, world)
        token = [sphx.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(token, src=0)
        group = sphx.SlabGroup(P, fluid, boundary, world, first_rank=rank, local_ranks=1, rccl_id=token[0], flags=flags)
        slabs_here, transport = 1, "RCCL point-to-point (ncclSend/ncclRecv grouped per exchange) over xGMI"
    else:
        slabs_here = max(1, int(getattr(args, "slabs", 1)))
        if getattr(args, "slab_transport", "loopback") == "rccl":      # the installed RCCL on one device: every message a send to self
            group = sphx.SlabGroup(P, fluid, boundary, slabs_here, first_rank=0, local_ranks=slabs_here, rccl_id=sphx.rccl_unique_id(), flags=flags)
            transport = "RCCL sends to self (%d slabs on one device, one-rank communicator)" % slabs_here
        else:
            group = sphx.SlabGroup(P, fluid, boundary, slabs_here, flags=flags)
            transport = "loopback (%d slabs on one device)" % slabs_here
    n_total = len(fluid)
    group.step(1)                                # = the constructor step of the single-device path
    if P.solver == sphx.PBD:
        group.step(1)                            # PBD: the constructor step only records positions
    if args.warmup > 0:
        group.step(args.warmup)
    # live roofline leg (rank 0's slab): hipEvents around every launch of the dominant kernel on the engine stream
    span = "density_error"
    if P.solver == sphx.DFSPH:
        sphx.kernel_timer(True, span)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    comm0 = group.comm_info()
    if world > 1 and comm0["transport"] == "rccl" and comm0["ranks"] != world:
        # (-1 = the library does not export ncclCommCount; anything else must be the launcher's world size)
        if comm0["ranks"] != -1:
            raise SystemExit("rank %d: the RCCL communicator reports %d ranks, the launcher %d" % (rank, comm0["ranks"], world))
    wait0 = group.wait_seconds()
    barrier()
    t0 = time.perf_counter()
    group.step(args.steps)
    torch.cuda.synchronize()
    wall_local = time.perf_counter() - t0
    barrier()
    wall = wall_local
    if world > 1:
        w = torch.tensor([wall_local], dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    if P.solver == sphx.DFSPH:
        bpp = 420 + 92 * args.div_iters + 104 * args.den_iters + 72 - (44 if args.div_iters >= 1 else 0)      # bench.py step_bytes_per_particle
        what = "DFSPH(%d div + %d density iters, fixed)" % (args.div_iters, args.den_iters)
    elif P.solver == sphx.WCSPH:
        bpp, what = 396, "WCSPH"
    else:
        bpp, what = 300 + 104 * P.pbd_iters + 72, "PBD(%d Jacobi iters)" % P.pbd_iters
    steps_per_s = args.steps / wall
    infos = [group.info(i) for i in range(slabs_here)]
    comm1 = group.comm_info()

    # this rank's own record: what it ran on, what it owned, how long IT took, what it sent -- and its own roofline leg
    roof = None
    if P.solver == sphx.DFSPH:
        spans = sphx.kernel_timer_collect()
        sphx.kernel_timer(False)
        if span in spans and spans[span][1] > 0:
            tot_ms, launches = spans[span]
            # a stage is launched on the edge layers of the sides that have a neighbour (ONE launch) and on the interior: one logical
            # launch = 2 timed spans per slab; a slab on its own (and the stage-then-exchange schedule) sweeps its particles in one
            alone = world == 1 and slabs_here == 1
            logical = launches / (1.0 if (flags or alone) else 2.0)   # logical launches of ALL local slabs; each works on `owned` particles
            avg_ms = tot_ms / logical
            owned = sum(o for _, _, o, _ in infos) / float(slabs_here)
            per_launch = 44.0 * owned                                   # bytes: bench.py RATE_KERNEL_BYTES_PER_PARTICLE
            achieved = per_launch / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_rate_quad<DENSITY_MODE> (span '%s'), this rank's slab(s), owned particles" % span,
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                    "avg_launch_ms": avg_ms, "launches": logical, "algorithmic_bytes_per_launch": per_launch}
    mine = {"rank": rank, "local_rank": local_rank, "device_pci_id": sphx.device_pci_id(local_rank),
            "device_name": torch.cuda.get_device_name(local_rank),
            "slabs": [{"columns": [a, b], "owned": o, "held": h} for a, b, o, h in infos],
            "ms_per_step": wall_local * 1e3 / args.steps, "host_wait_seconds": group.wait_seconds() - wait0,
            "rccl": {"transport": comm1["transport"], "comm_ranks": comm1["ranks"], "comm_rank": comm1["rank"]},
            "halo_bytes_sent_per_step": (comm1["bytes_sent"] - comm0["bytes_sent"]) / float(args.steps),
            "halo_bytes_received_per_step": (comm1["bytes_received"] - comm0["bytes_received"]) / float(args.steps),
            "exchanges_per_step": (comm1["exchanges"] - comm0["exchanges"]) / float(args.steps),
            "allreduces_per_step": (comm1["allreduces"] - comm0["allreduces"]) / float(args.steps),
            "roofline": roof}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    ms = [r["ms_per_step"] for r in per_rank]
    owned_all = [sum(sl["owned"] for sl in r["slabs"]) for r in per_rank]
    # The base of this line's scaling figure, measured in the SAME run: the whole workload as ONE slab of the same layer in the same
    # arithmetic on rank 0's device (the plain single-device engine -- bench.py --gpus 1 -- is another code path: one captured graph,
    # rows that persist; it is not the base of a slab curve).  Skipped with --no-extra-legs.
    scaling_base = None
    if (world > 1 or slabs_here > 1) and not getattr(args, "no_extra_legs", False):
        if rank == 0:
            one = sphx.SlabGroup(P, fluid, boundary, 1, flags=flags)
            one.step(1 + max(args.warmup, 1))
            torch.cuda.synchronize()
            k = max(1, min(args.steps, 20))
            tb = time.perf_counter(); one.step(k); torch.cuda.synchronize()
            base_ms = (time.perf_counter() - tb) * 1e3 / k
            one.close()
            scaling_base = {"what": "the same workload as ONE slab of the slab layer (loopback) on rank 0's device, %s arithmetic" % arith_ran,
                            "ms_per_step": base_ms, "steps_per_s": 1e3 / base_ms, "steps": k,
                            "speedup_of_this_line": base_ms / (wall * 1e3 / args.steps)}
        if world > 1:
            dist.barrier()
    result = {
        "metric": "simulation steps/sec, %s dam-break" % ("DFSPH" if P.solver == sphx.DFSPH else solver_name), "value": steps_per_s,
        "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "dam-break %dx%dx%d = %d fluid + %d boundary particles, %s, dt=%g, %s arithmetic"
                               % (args.nx, 3 * args.nx // 2, args.nx, n_total, len(boundary), what, P.dt,
                                  "strict (bit-exact IEEE)" if arith_ran == "strict" else "tolerance (1e-5 contract; rows rebuilt every step)"),
                   "arithmetic": arith_ran, "arithmetic_asked": arith_asked,
                   "particles": n_total,
                   "decomposition": "%d x-slabs; transport %s; %s" % (world if world > 1 else slabs_here, transport,
                                    "stage-then-exchange" if flags else "edge-first stages, halo overlapped with interior sweeps"),
                   "step_algorithmic_bytes_per_particle": bpp,
                   "step_algorithmic_GBps": bpp * n_total * steps_per_s / 1e9,
                   "step_hbm_roofline_frac_of_job": bpp * n_total * steps_per_s / 1e9 / (8000.0 * world)},
        # self-verification of a multi-process run: what RCCL itself says about the communicator, which physical device every
        # rank used, and every rank's own clock, ownership, traffic and roofline leg
        "rccl": {"ranks": comm1["ranks"], "transport": comm1["transport"], "world_size_launcher": world,
                 "distinct_devices": len({r["device_pci_id"] for r in per_rank}) if world > 1 else 1},
        "ranks": per_rank,
        "rank_ms_per_step": {"min": min(ms), "max": max(ms)},
        "owned_particles": {"min": min(owned_all), "max": max(owned_all), "total": sum(owned_all)},
        "roofline": per_rank[0]["roofline"],
        "scaling_base": scaling_base,
    }
    group.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result
