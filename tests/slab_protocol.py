"""slab_protocol.py — TEST-SIDE restatement of the x-slab protocol over torch.distributed (moved out of the product package in r03).

The product is the native layer csrc/slab.hip; this file is the same protocol written against torch.distributed and an
abstract engine so that the N > 1 logic can run under gloo on machines without a GPU (tests/test_slab_cpu.py, with
tests/slab_cpu_engine.py) and, with HipSlabEngine, against the HIP engine (tests/test_gpu_slab.py).


One process per GPU.  The linear cell id runs x slowest, so rank r's slab of cell columns
[x0, x1) plus its one-cell halos x0-1 and x1 is ONE contiguous range of the globally cell-sorted
particle arrays.  Every rank therefore holds a local array [left ghosts | owned | right ghosts]
that is, entry for entry, a slice of the array a single device would hold; every per-particle sum
visits the same neighbours in the same order, so the distributed run is bit-identical to the
single-device run for any number of ranks (tests/test_slab_*.py check exactly that).

Per step (fixed DFSPH iteration counts v, d):
  1. particle exchange: every rank sends each neighbour the particles it owned last step whose new
     cell column lies within one column of the shared cut (migrants and ghost copies alike, in
     array order; pos, vel, id, warm stiffness = 32 B each) and rebuilds its pre-sort array as
     [from left | previously owned | from right] — ascending in last step's global order, which is
     what keeps the stable cell sort identical to the single-device one;
  2. the engine's stages (sphx_run_phase) with a halo refresh after each stage that writes a field
     the next stage reads from neighbours (kappa, vel4, cg4).  Because the boundary layers are
     contiguous ranges of the sorted arrays, a halo message is a plain slice: no pack kernels.
     v + d + 3 refreshes of the velocity mirror, v + d + 1 of kappa, one of the colour gradient.

Collectives: only neighbour point-to-point messages (dist.batch_isend_irecv, i.e. grouped
ncclSend/ncclRecv over xGMI on GPUs); no all-reduce in fixed-iteration mode.  The engine enqueues on
torch's current stream, so NCCL's stream dependencies order messages with kernels without host
synchronisation; the only host round trips per step are the message sizes and the five layer offsets.

DFSPH runs with fixed iteration counts or adaptively (the reference's loops): the
termination sum is an exact integer, each rank sums its owned particles and a 1-word
all-reduce gives every rank the identical total, so iteration counts equal the single-device ones.
WCSPH uses four stages and two halo refreshes (colour gradient, pressure term).
PBD sweeps run on positions that moved after binning: a particle binned in an edge
column may sit one column further out when it is swept, so PBD slabs keep TWO ghost columns per side
(still one contiguous range) and refresh lambda and the position mirror inside every Jacobi iteration,
then the velocity mirror and the colour gradient: 2 * iters + 2 refreshes per step.

Two host drivers implement this protocol.  The product path is the native layer csrc/slab.hip (C++, RCCL or
loopback transport, device-side particle exchange, edge-first stages so that halo traffic overlaps the interior
sweeps): run_slab_bench() below only bootstraps it (RCCL token, barriers).  The classes in this file are the
protocol written against torch.distributed and an abstract engine: with the CPU stand-in engine of
tests/slab_cpu_engine.py they exercise the N > 1 protocol under gloo on machines without a GPU
(tests/test_slab_cpu.py), with HipSlabEngine they drive the HIP engine over gloo or RCCL.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

(PH_SEARCH, PH_HEAD, PH_DIV_CORRECT, PH_DIV_ERROR, PH_FORCE, PH_VISC_COLOR, PH_SURFACE, PH_WARM_CORRECT,
 PH_DEN_ERROR_SET, PH_DEN_CORRECT, PH_DEN_ERROR_ACC, PH_ADVECT, PH_W_SEARCH, PH_W_PROPS, PH_W_SURFACE,
 PH_W_PRESSURE, PH_P_SEARCH, PH_P_LAMBDA, PH_P_DELTA, PH_P_VELOCITY, PH_P_XSPH, PH_P_SURFACE, PH_P_TAIL) = range(23)

EPS = 1e-6


# ------------------------------------------------------------------------------------ partitioning
def cell_column(x, cell_length):
    """global cell column of positions x (float32): true fp32 division, truncation"""
    if isinstance(x, torch.Tensor):
        return torch.div(x, torch.tensor(cell_length, dtype=torch.float32, device=x.device)).to(torch.int32)
    return (x.astype(np.float32) / np.float32(cell_length)).astype(np.int32)


def choose_cuts(columns, gx, world, min_width=2):
    """cut planes x_0=0 < x_1 < ... < x_world=gx balancing particle counts; every slab >= min_width columns
    (ghost width + 1: a particle that moves one column must be deliverable by its last owner to every
    rank that needs it, owner or ghost holder, with neighbour messages only)"""
    hist = np.bincount(np.clip(columns, 0, gx - 1), minlength=gx).astype(np.int64)
    cdf = np.cumsum(hist)
    total = int(cdf[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        xc = int(np.searchsorted(cdf, target, side="left"))      # the column in which the running count crosses the target
        below = float(cdf[xc - 1]) if xc > 0 else 0.0
        above = float(cdf[xc]) if xc < gx else float(total)
        x = xc if (target - below < above - target) else xc + 1   # ... goes to the side that leaves the smaller error
        x = max(x, cuts[-1] + min_width)
        x = min(x, gx - min_width * (world - r))
        cuts.append(x)
    cuts.append(gx)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b - a < min_width:
            raise ValueError("domain too narrow for %d slabs: cuts %s" % (world, cuts))
    return cuts


# ------------------------------------------------------------------------------------ transport
class Neighbors:
    """point-to-point exchange with the left/right slab neighbour (None at the domain ends)"""

    def __init__(self, rank, world):
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.stage_through_host = world > 1 and dist.get_backend() == "gloo"

    def allreduce_int(self, value, device):
        """sum of one Python int over all ranks (exact: int64)"""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return int(value)
        dev = "cpu" if dist.get_backend() == "gloo" else device
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def exchange(self, to_left, to_right, from_left, from_right):
        """send to_left/to_right (tensors or None), receive into from_left/from_right (tensors or None)"""
        ops, staged = [], []

        # Device tensors handed to NCCL are always torch-allocated copies: the engine's arrays are
        # foreign memory to torch's caching allocator (zero-copy views), and a halo slice is small.
        def prep_send(t):
            if t is None or t.numel() == 0:
                return None
            if t.is_cuda:
                return t.contiguous().cpu() if self.stage_through_host else t.clone(memory_format=torch.contiguous_format)
            return t.contiguous()

        def prep_recv(t):
            if t is None or t.numel() == 0:
                return None
            if t.is_cuda:
                h = torch.empty(t.shape, dtype=t.dtype, device="cpu" if self.stage_through_host else t.device)
                staged.append((t, h))
                return h
            if not t.is_contiguous():
                h = torch.empty_like(t)
                staged.append((t, h))
                return h
            return t

        for peer, s, r in ((self.left, to_left, from_left), (self.right, to_right, from_right)):
            if peer is None:
                continue
            s, r = prep_send(s), prep_recv(r)
            if s is not None:
                ops.append(dist.P2POp(dist.isend, s, peer))
            if r is not None:
                ops.append(dist.P2POp(dist.irecv, r, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, h in staged:
            dst.copy_(h)


# ------------------------------------------------------------------------------------ engines
class _DevView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


class HipSlabEngine:
    """the HIP engine (libsphx.so) on one slab; fields are zero-copy torch views of device memory"""
    zero_copy = True

    def __init__(self, sphx, params, cap, boundary_pos, boundary_mass, device):
        self.sphx, self.cap, self.device = sphx, cap, device
        self.count = cap
        self.sys = sphx.System(params, np.zeros((cap, 3), np.float32), boundary_pos, ctor_step=False)
        if len(boundary_pos):
            self.sys.set(sphx.F_BMASS, np.ascontiguousarray(boundary_mass, np.float32))
        self.C = self.sys.cells

        def view(field, comps, typestr="<f4"):
            shape = (cap, comps) if comps > 1 else (cap,)
            return torch.as_tensor(_DevView(self.sys.device_ptr(field), shape, typestr), device=device)

        self.f = {"pos": view(sphx.F_POS, 3), "vel": view(sphx.F_VEL, 3), "ids": view(sphx.F_ID, 1, "<i4"),
                  "vel_nbr": view(sphx.F_VEL4, 4), "cg_nbr": view(sphx.F_CG4, 4), "density": view(sphx.F_DENSITY, 1),
                  "pterm": view(sphx.F_PTERM, 1),
                  "posf": view(sphx.F_POSF, 4)}      # (x, y, z, scalar): what the one-gather sweeps read from neighbours
        if params.solver == sphx.DFSPH:
            self.f["warm"] = view(sphx.F_WARM, 1)
            self.f["kappa"] = view(sphx.F_KAPPA, 1)
        if params.solver == sphx.PBD:
            self.f["pos_last"] = view(sphx.F_POS_LAST, 3)
            self.f["lambda"] = view(sphx.F_LAMBDA, 1)
            self.f["pos_nbr"] = view(sphx.F_POS4, 4)      # (x, y, z, mass): what the sweeps gather
        self.pressure_halo = ["pterm"]     # what the pressure-force stage reads from neighbours
        self.cell_start = torch.as_tensor(_DevView(self.sys.device_ptr(sphx.F_CELLSTART_F), (self.C + 1,), "<i4"),
                                          device=device)

    def set_count(self, n):
        self.sys.set_count(n)
        self.count = n

    def run(self, phase):
        self.sys.run_phase(phase)

    def run_reduce(self, phase, lo, hi):
        """error stage with the exact |error| sum over particles [lo, hi); returns the integer"""
        self.sys.run_phase_reduce(phase, lo, hi)
        return self.sys.error_total_fixed()

    def has(self, name):
        return name in self.f

    def read(self, name, lo, hi):
        return self.f[name][lo:hi]

    def write(self, name, lo, t):
        self.f[name][lo:lo + t.shape[0]].copy_(t)

    def cell_starts(self, idx):
        return [int(v) for v in self.cell_start[torch.as_tensor(idx, device=self.device, dtype=torch.long)].cpu()]

    def columns(self, lo, hi, cell_length):
        """global cell column of particles [lo, hi), computed by the engine's own division"""
        out = torch.empty(hi - lo, dtype=torch.int32, device=self.device)
        if hi > lo:
            self.sphx.cell_columns(self.f["pos"][lo:hi].data_ptr(), hi - lo, cell_length, out.data_ptr())
        return out

    def to_device(self, arr):
        return torch.as_tensor(arr, device=self.device)


# ------------------------------------------------------------------------------------ the driver
class SlabDriver:
    def __init__(self, engine, nbrs, x0, x1, gy, gz, cell_length, div_iters, den_iters, surface, timers=None,
                 solver="dfsph", adaptive=None, ghost=1, pbd_iters=0):
        """adaptive: None, or dict(n_global, rho0, div_thr, den_thr, max_iter) for the reference's loops;
        ghost: ghost cell columns per side (1; PBD: 2, its sweeps run on positions that moved after binning)"""
        self.solver, self.adaptive = solver, adaptive
        self.g, self.pbd_iters, self.steps_done = ghost, pbd_iters, 0
        # per-particle state that travels with a particle besides pos, vel, id
        self.extras = {"dfsph": [("warm", 1)], "pbd": [("pos_last", 3)]}.get(solver, [])
        self.iters = (0, 0)
        self.e, self.nb = engine, nbrs
        self.x0, self.x1, self.L = x0, x1, gy * gz
        self.gxl = (x1 - x0) + 2 * ghost
        self.cl = cell_length
        self.v, self.d, self.surface = div_iters, den_iters, surface
        self.owned = (0, 0)
        self.layers = None
        self.t_comm = 0.0
        self.timers = timers

    # -- initial distribution: this rank's owned particles in generation order
    def load_initial(self, pos, vel=None):
        n = pos.shape[0]
        e = self.e
        if n > e.cap:
            raise RuntimeError("slab capacity %d too small for %d particles" % (e.cap, n))
        e.write("pos", 0, pos)
        e.write("vel", 0, vel if vel is not None else torch.zeros_like(pos))
        if e.has("warm"):
            e.write("warm", 0, torch.zeros(n, dtype=torch.float32, device=pos.device))
        if e.has("pos_last"):
            e.write("pos_last", 0, pos)        # PBDSolver.h:56-60: the first step records the positions
        self.owned = (0, n)

    def _exchange_particles(self):
        e, g = self.e, self.g
        o0, o1 = self.owned
        pos, vel = e.read("pos", o0, o1), e.read("vel", o0, o1)
        ids = e.read("ids", o0, o1)
        cols = [pos, vel, ids.view(torch.float32).unsqueeze(1)]
        for name, comps in self.extras:
            t = e.read(name, o0, o1)
            cols.append(t if comps > 1 else t.unsqueeze(1))
        col = e.columns(o0, o1, self.cl)
        if o1 > o0 and (int(col.min()) < self.x0 - 1 or int(col.max()) > self.x1):
            raise RuntimeError("a particle crossed more than one cell column in one step")
        payload = torch.cat(cols, dim=1)                                   # [m, 7 + extras]
        width = payload.shape[1]
        to_left = payload[col <= self.x0 + g - 1] if self.nb.left is not None else None
        to_right = payload[col >= self.x1 - g] if self.nb.right is not None else None
        dev = payload.device
        # message sizes first
        cnt_send_l = torch.tensor([0 if to_left is None else to_left.shape[0]], dtype=torch.int64, device=dev)
        cnt_send_r = torch.tensor([0 if to_right is None else to_right.shape[0]], dtype=torch.int64, device=dev)
        cnt_recv_l = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt_recv_r = torch.zeros(1, dtype=torch.int64, device=dev)
        self.nb.exchange(cnt_send_l, cnt_send_r, cnt_recv_l, cnt_recv_r)
        nl, nr = int(cnt_recv_l.item()), int(cnt_recv_r.item())
        from_left = torch.empty((nl, width), dtype=torch.float32, device=dev)
        from_right = torch.empty((nr, width), dtype=torch.float32, device=dev)
        self.nb.exchange(to_left, to_right, from_left, from_right)
        pre = torch.cat([from_left, payload, from_right], dim=0)
        n = pre.shape[0]
        if n > e.cap:
            raise RuntimeError("slab capacity %d exceeded (%d particles)" % (e.cap, n))
        e.write("pos", 0, pre[:, 0:3])
        e.write("vel", 0, pre[:, 3:6])
        e.write("ids", 0, pre[:, 6].contiguous().view(torch.int32))
        at = 7
        for name, comps in self.extras:
            e.write(name, 0, pre[:, at:at + comps] if comps > 1 else pre[:, at].contiguous())
            at += comps
        e.set_count(n)

    def _update_layers(self):
        L, g, w = self.L, self.gxl, self.g
        c = self.e.cell_starts([w * L, 2 * w * L, (g - 2 * w) * L, (g - w) * L, g * L])
        # [0,c0) left ghosts, [c0,c1) first w owned layers, [c2,c3) last w owned layers, [c3,c4) right ghosts
        self.layers = c
        self.owned = (c[0], c[3])

    def _halo(self, name):
        # scalars that one-gather sweeps read through the packed (x, y, z, scalar) records travel in that array too
        if name in ("kappa", "pterm", "lambda") and self.e.has("posf"):
            self._halo("posf")
        t0 = time.perf_counter() if self.timers is not None else 0.0
        c0, c1, c2, c3, c4 = self.layers
        e = self.e
        send_l = e.read(name, c0, c1) if self.nb.left is not None else None
        send_r = e.read(name, c2, c3) if self.nb.right is not None else None
        recv_l = e.read(name, 0, c0) if self.nb.left is not None else None
        recv_r = e.read(name, c3, c4) if self.nb.right is not None else None
        copies = not getattr(e, "zero_copy", False)      # engines whose read() returns copies need a write-back
        if copies:
            if recv_l is not None:
                recv_l = torch.empty_like(recv_l)
            if recv_r is not None:
                recv_r = torch.empty_like(recv_r)
        self.nb.exchange(send_l, send_r, recv_l, recv_r)
        if copies:
            if recv_l is not None and recv_l.numel():
                e.write(name, 0, recv_l)
            if recv_r is not None and recv_r.numel():
                e.write(name, c3, recv_r)
        if self.timers is not None:
            self.timers["halo"] = self.timers.get("halo", 0.0) + time.perf_counter() - t0

    def _global_error(self, phase):
        """run an error stage, all-reduce the exact integer |error| sum of the owned particles and return
        it as the fp32 value DFSPHSolver compares with its threshold"""
        fixed = self.e.run_reduce(phase, self.owned[0], self.owned[1])
        total = self.nb.allreduce_int(fixed, getattr(self.e, "device", "cpu"))
        return np.float32(np.float64(total) * (1.0 / 4294967296.0))

    def _step_wcsph(self):
        e = self.e
        self._exchange_particles()
        e.run(PH_W_SEARCH)
        self._update_layers()
        e.run(PH_W_PROPS)
        if self.surface:
            self._halo("cg_nbr")
        for name in e.pressure_halo:
            self._halo(name)
        e.run(PH_W_SURFACE)
        e.run(PH_W_PRESSURE)
        e.run(PH_ADVECT)

    def _step_pbd(self):
        """PBDSolver::step (PBDSolver.cu:34-79).  The k-th call equals the k-th SPHSystem::step() of the
        single-device system counting its constructor step, which for PBD only sorts the particles and
        records their positions (PBDSolver.cu:45-49)."""
        e = self.e
        self._exchange_particles()
        e.run(PH_P_SEARCH)
        self._update_layers()
        self.steps_done += 1
        if self.steps_done == 1:
            return
        for _ in range(self.pbd_iters):
            e.run(PH_P_LAMBDA); self._halo("lambda")
            e.run(PH_P_DELTA); self._halo("pos_nbr")
        e.run(PH_P_VELOCITY); self._halo("vel_nbr")
        e.run(PH_P_XSPH)
        if self.surface:
            self._halo("cg_nbr")
        e.run(PH_P_SURFACE)
        e.run(PH_P_TAIL)

    def step(self):
        if self.solver == "wcsph":
            return self._step_wcsph()
        if self.solver == "pbd":
            return self._step_pbd()
        e = self.e
        ad = self.adaptive
        self._exchange_particles()
        e.run(PH_SEARCH)
        self._update_layers()
        e.run(PH_HEAD); self._halo("kappa")
        if ad is None:
            for _ in range(self.v):
                e.run(PH_DIV_CORRECT); self._halo("vel_nbr")
                e.run(PH_DIV_ERROR); self._halo("kappa")
            it_div = self.v
        else:       # DFSPHSolver.cu:347-361
            limit = np.float32(ad["div_thr"]) * np.float32(ad["n_global"]) * np.float32(ad["rho0"])
            it_div, total = 0, np.float32(3.4028235e38)
            while (it_div < 1 or total > limit) and it_div < ad["max_iter"]:
                e.run(PH_DIV_CORRECT); self._halo("vel_nbr")
                total = self._global_error(PH_DIV_ERROR); self._halo("kappa")
                it_div += 1
        e.run(PH_FORCE)
        e.run(PH_VISC_COLOR)
        if self.surface:
            self._halo("cg_nbr")
        e.run(PH_SURFACE); self._halo("vel_nbr")
        e.run(PH_WARM_CORRECT); self._halo("vel_nbr")
        e.run(PH_DEN_ERROR_SET); self._halo("kappa")
        if ad is None:
            for k in range(self.d):
                e.run(PH_DEN_CORRECT); self._halo("vel_nbr")
                e.run(PH_DEN_ERROR_ACC)
                if k + 1 < self.d:
                    self._halo("kappa")
            it_den = self.d
        else:       # DFSPHSolver.cu:187-208
            limit = np.float32(ad["den_thr"]) * np.float32(ad["n_global"]) * np.float32(ad["rho0"])
            it_den, total = 0, np.float32(3.4028235e38)
            while (it_den < 2 or total > limit) and it_den < ad["max_iter"]:
                e.run(PH_DEN_CORRECT); self._halo("vel_nbr")
                it_den += 1
                if it_den >= 2:
                    total = self._global_error(PH_DEN_ERROR_ACC)
                else:
                    e.run(PH_DEN_ERROR_ACC)
                self._halo("kappa")
        self.iters = (it_div, it_den)
        e.run(PH_ADVECT)

    def owned_state(self):
        """(ids, pos, vel, density) of the particles this rank owns, as CPU numpy arrays"""
        o0, o1 = self.owned
        e = self.e
        return tuple(e.read(k, o0, o1).cpu().numpy().copy() for k in ("ids", "pos", "vel", "density"))


# ------------------------------------------------------------------------------------ set-up helper
def build_slab(make_engine, scene_params, fluid, boundary_sorted, boundary_mass, rank, world, capacity_factor=1.3,
               velocity=None):
    """cuts the global scene into `world` x-slabs and creates this rank's engine + driver.
    `boundary_sorted`/`boundary_mass`: the GLOBAL boundary set in cell-sorted order with its masses
    (computed by a whole-domain boundary-only system, SPHSystem.cu:69-71).  `make_engine(params,
    cap, bpos, bmass)` returns the engine (HipSlabEngine; the CPU tests plug in a stand-in with the same interface)."""
    P = scene_params
    gx, gy, gz = P.cells[0], P.cells[1], P.cells[2]
    cl = P.cell_length
    solver = {0: "wcsph", 1: "dfsph", 2: "pbd"}[P.solver]
    ghost = 2 if solver == "pbd" else 1
    col = cell_column(fluid[:, 0], cl)
    cuts = choose_cuts(col, gx, world, min_width=ghost + 1)
    x0, x1 = cuts[rank], cuts[rank + 1]
    mine = fluid[(col >= x0) & (col < x1)]
    bcol = cell_column(boundary_sorted[:, 0], cl)
    bsel = (bcol >= x0 - ghost) & (bcol <= x1 + ghost - 1)
    counts = [int(((col >= a) & (col < b)).sum()) for a, b in zip(cuts[:-1], cuts[1:])]
    cap = int(max(counts) * capacity_factor) + 4096
    Pl = type(P)()
    for name, _ in P._fields_:
        setattr(Pl, name, getattr(P, name))
    Pl.cells[0] = (x1 - x0) + 2 * ghost
    Pl.reserved[1] = x0 - ghost      # global column of local column 0
    Pl.reserved[2] = 1               # slab system (also when the offset is 0)
    engine = make_engine(Pl, cap, np.ascontiguousarray(boundary_sorted[bsel]), np.ascontiguousarray(boundary_mass[bsel]))
    surface = P.surface_tension > EPS or P.air_pressure > EPS
    nbrs = Neighbors(rank, world)
    adaptive = None
    if solver == "dfsph" and (P.dfsph_fixed_div < 0 or P.dfsph_fixed_den < 0):
        adaptive = dict(n_global=len(fluid), rho0=P.rho0, div_thr=P.dfsph_divergence_thr, den_thr=P.dfsph_density_thr,
                        max_iter=P.dfsph_max_iter)
    drv = SlabDriver(engine, nbrs, x0, x1, gy, gz, cl, P.dfsph_fixed_div, P.dfsph_fixed_den, surface, solver=solver,
                     adaptive=adaptive, ghost=ghost, pbd_iters=P.pbd_iters)
    sel = (col >= x0) & (col < x1)
    drv.load_initial(engine.to_device(np.ascontiguousarray(mine)),
                     None if velocity is None else engine.to_device(np.ascontiguousarray(velocity[sel])))
    # ids: global generation index of each initial particle
    gid = np.flatnonzero((col >= x0) & (col < x1)).astype(np.int32)
    engine.write("ids", 0, engine.to_device(gid))
    return drv, cuts, counts


def engine_count(drv):
    """particles the engine currently sweeps over (owned + ghost copies)"""
    return int(drv.e.count)

