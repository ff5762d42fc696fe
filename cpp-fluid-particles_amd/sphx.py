"""ctypes binding of libsphx.so (include/sphx_c.h) — the host-side Python view of the engine.

This is plumbing for tests, bench.py and the multi-GPU driver: every call goes straight through
the C ABI into the HIP engine.  There is no CPU path; without the built library or without a HIP
device the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPHX_LIB", os.path.join(_HERE, "libsphx.so"))

WCSPH, DFSPH, PBD = 0, 1, 2

(F_POS, F_VEL, F_DENSITY, F_PRESSURE, F_MASS, F_CELL, F_CELLSTART_F, F_CELLSTART_B, F_ID, F_BPOS,
 F_BMASS, F_ALPHA, F_KAPPA, F_ERROR, F_WARM, F_POS_LAST, F_LAMBDA, F_BUF3, F_VEL4, F_CG4, F_PTERM, F_POS4, F_POSF) = range(23)

(PH_SEARCH, PH_HEAD, PH_DIV_CORRECT, PH_DIV_ERROR, PH_FORCE, PH_VISC_COLOR, PH_SURFACE, PH_WARM_CORRECT,
 PH_DEN_ERROR_SET, PH_DEN_CORRECT, PH_DEN_ERROR_ACC, PH_ADVECT, PH_W_SEARCH, PH_W_PROPS, PH_W_SURFACE,
 PH_W_PRESSURE, PH_P_SEARCH, PH_P_LAMBDA, PH_P_DELTA, PH_P_VELOCITY, PH_P_XSPH, PH_P_SURFACE, PH_P_TAIL,
 PH_SURFACE_WARM, PH_W_SURFACE_PRESSURE, PH_P_DELTA_SWEEP, PH_P_APPLY) = range(27)

_INT_FIELDS = (F_CELL, F_CELLSTART_F, F_CELLSTART_B, F_ID)
_VEC_FIELDS = (F_POS, F_VEL, F_BPOS, F_POS_LAST, F_BUF3)

EXPORTS = [
    "sphx_last_error", "sphx_device_count", "sphx_set_device", "sphx_sizeof_params",
    "sphx_scene_params", "sphx_scene_counts", "sphx_scene_fill", "sphx_create", "sphx_destroy",
    "sphx_step", "sphx_step_n", "sphx_counts", "sphx_iters", "sphx_field_bytes", "sphx_get",
    "sphx_set", "sphx_device_ptr", "sphx_profile_step", "sphx_eval_kernels", "sphx_ieee_probe",
    "sphx_generate_dots", "sphx_kernel_timer", "sphx_kernel_timer_collect", "sphx_run_phase", "sphx_run_phase_reduce", "sphx_error_total_fixed", "sphx_set_count", "sphx_use_stream",
    "sphx_sync", "sphx_cell_columns", "sphx_fastmath_selftest", "sphx_snapshot_save", "sphx_snapshot_load",
    "sphx_tuning_defaults", "sphx_set_tuning", "sphx_get_tuning", "sphx_invalidate_order", "sphx_last_rate_kernel",
    "sphx_get_params", "sphx_row_stats", "sphx_row_walk_stats", "sphx_row_capacity", "sphx_rows_partial", "sphx_rows_stale", "sphx_persistent_stats", "sphx_device_pci_id", "sphx_sample_box", "sphx_sample_sphere", "sphx_sample_triangles",
]
# symbols exported under the reference's own names (vbo.cu:46-51)
REFERENCE_EXPORTS = ["generate_dots"]


class Params(C.Structure):
    """sphx_params (include/sphx_c.h)."""
    _fields_ = [
        ("space", C.c_float * 3), ("cells", C.c_int * 3),
        ("cell_length", C.c_float), ("radius", C.c_float), ("dt", C.c_float), ("m0", C.c_float),
        ("rho0", C.c_float), ("rho_boundary", C.c_float), ("stiff", C.c_float), ("visc", C.c_float),
        ("surface_tension", C.c_float), ("air_pressure", C.c_float), ("gravity", C.c_float * 3),
        ("solver", C.c_int),
        ("dfsph_density_thr", C.c_float), ("dfsph_divergence_thr", C.c_float),
        ("dfsph_max_iter", C.c_int), ("dfsph_fixed_div", C.c_int), ("dfsph_fixed_den", C.c_int),
        ("pbd_iters", C.c_int), ("pbd_xsph_c", C.c_float), ("pbd_relaxation", C.c_float),
        ("pow7_mode", C.c_int), ("xsph_mode", C.c_int), ("reserved", C.c_int * 4),
    ]

    def copy(self):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class Tuning(C.Structure):
    """sphx_tuning (include/sphx_c.h): the engine's behaviour switches, installed process-wide with set_tuning()."""
    _fields_ = [
        ("struct_size", C.c_int), ("engine_flags", C.c_int), ("row_capacity", C.c_int), ("quad_mask", C.c_int), ("duo_mask", C.c_int),
        ("quad_mask_tol", C.c_int), ("tol_strict_rate", C.c_int), ("brick", C.c_int), ("brick_min", C.c_int), ("range_order", C.c_int),
        ("range_order_min", C.c_int), ("force_tile_order", C.c_int), ("no_fastmath", C.c_int), ("no_graph", C.c_int), ("graph_debug", C.c_int),
        ("dfsph_host_loop", C.c_int), ("dfsph_window", C.c_int), ("dfsph_no_tail", C.c_int), ("no_kick_fusion", C.c_int),
        ("pbd_skin", C.c_float), ("pbd_skin_fixed", C.c_int), ("persist_controller", C.c_int), ("slab_edge_stream", C.c_int),
        ("slab_comm_priority", C.c_int), ("dfsph_tail_flat", C.c_int), ("group_build_max", C.c_int), ("pbd_no_partial", C.c_int), ("reserved", C.c_int * 5),
    ]


class SphxError(RuntimeError):
    pass


def build(force=False):
    """Compile libsphx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", _HERE, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SphxError("libsphx.so is not built (run `make -C cpp-fluid-particles_amd` or "
                            "__graft_entry__.build()); the engine has no fallback path")
        L = C.CDLL(LIB_PATH)
        L.sphx_last_error.restype = C.c_char_p
        L.sphx_scene_params.argtypes = [C.c_int, C.POINTER(Params)]
        L.sphx_scene_counts.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sphx_scene_fill.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.sphx_create.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_void_p)]
        L.sphx_destroy.argtypes = [C.c_void_p]
        L.sphx_step.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.sphx_step_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.sphx_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        L.sphx_iters.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 2
        L.sphx_field_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.sphx_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.sphx_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.sphx_device_ptr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.sphx_profile_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.sphx_eval_kernels.argtypes = [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4
        L.sphx_ieee_probe.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
        L.sphx_generate_dots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sphx_run_phase.argtypes = [C.c_void_p, C.c_int]
        L.sphx_set_count.argtypes = [C.c_void_p, C.c_int]
        L.sphx_run_phase_reduce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.sphx_error_total_fixed.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.sphx_use_stream.argtypes = [C.c_void_p]
        L.sphx_cell_columns.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.sphx_snapshot_save.argtypes = [C.c_void_p, C.c_char_p]
        L.sphx_snapshot_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.sphx_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.sphx_row_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_void_p]
        L.sphx_row_capacity.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.sphx_kernel_timer.argtypes = [C.c_int, C.c_char_p]
        L.sphx_kernel_timer_collect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.sphx_tuning_defaults.argtypes = [C.POINTER(Tuning)]
        L.sphx_set_tuning.argtypes = [C.POINTER(Tuning)]
        L.sphx_get_tuning.argtypes = [C.POINTER(Tuning)]
        L.sphx_invalidate_order.argtypes = [C.c_void_p]
        L.sphx_last_rate_kernel.argtypes = [C.c_char_p, C.c_int]
        if L.sphx_sizeof_params() != C.sizeof(Params):
            raise SphxError("sphx_params layout mismatch between sphx.py and libsphx.so")
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SphxError("sphx error %d: %s" % (rc, lib().sphx_last_error().decode()))


def device_count():
    return lib().sphx_device_count()


def default_tuning():
    t = Tuning()
    _check(lib().sphx_tuning_defaults(C.byref(t)))
    # the library writes ITS sizeof into struct_size: a layout drift between this binding and sphx_tuning must not pass silently
    if t.struct_size != C.sizeof(Tuning):
        raise SphxError("sphx_tuning layout mismatch between sphx.py (%d bytes) and libsphx.so (%d bytes)" % (C.sizeof(Tuning), t.struct_size))
    return t


def set_tuning(tuning=None, **fields):
    """install a tuning block process-wide (systems and slab groups created afterwards use it; the `live` fields act at once).
    set_tuning() restores the defaults; set_tuning(row_capacity=12, dfsph_no_tail=1) = the defaults with these fields changed."""
    if tuning is None and not fields:
        _check(lib().sphx_set_tuning(None))
        return
    t = tuning if tuning is not None else default_tuning()
    for k, v in fields.items():
        if not hasattr(t, k):
            raise SphxError("sphx_tuning has no field %r" % k)
        setattr(t, k, v)
    if tuning is None:
        t.struct_size = C.sizeof(Tuning)       # (a block the caller built keeps what it says about itself: the library checks it)
    _check(lib().sphx_set_tuning(C.byref(t)))


def get_tuning():
    t = Tuning()
    _check(lib().sphx_get_tuning(C.byref(t)))
    return t


def last_rate_kernel():
    """(variant number, name) of the kernel the most recent DFSPH error sweep was launched as"""
    buf = C.create_string_buffer(128)
    v = lib().sphx_last_rate_kernel(buf, 128)
    if v < 0:
        _check(v)
    return v, buf.value.decode()


def set_device(ordinal):
    _check(lib().sphx_set_device(ordinal))


def scene(nx):
    """(params, fluid[n,3], boundary[nb,3]) of the dam-break scene (nx=24: the reference scene)."""
    L = lib()
    P = Params()
    _check(L.sphx_scene_params(nx, C.byref(P)))
    n, nb = C.c_int(), C.c_int()
    _check(L.sphx_scene_counts(nx, C.byref(n), C.byref(nb)))
    fluid = np.empty((n.value, 3), np.float32)
    boundary = np.empty((nb.value, 3), np.float32)
    _check(L.sphx_scene_fill(nx, fluid.ctypes.data, boundary.ctypes.data))
    return P, fluid, boundary


class System:
    """One SPHSystem on the current HIP device."""

    def __init__(self, params, fluid, boundary, ctor_step=True):
        fluid = np.ascontiguousarray(fluid, np.float32).reshape(-1, 3)
        boundary = np.ascontiguousarray(boundary, np.float32).reshape(-1, 3)
        self.n, self.nb = len(fluid), len(boundary)
        self.cells = params.cells[0] * params.cells[1] * params.cells[2]
        self.params = params
        h = C.c_void_p()
        _check(lib().sphx_create(C.byref(params), fluid.ctypes.data, self.n, boundary.ctypes.data, self.nb,
                                 int(ctor_step), C.byref(h)))
        self._h = h

    def step(self):
        ms = C.c_float()
        _check(lib().sphx_step(self._h, C.byref(ms)))
        return ms.value

    def step_n(self, n):
        ms = C.c_float()
        _check(lib().sphx_step_n(self._h, n, C.byref(ms)))
        return ms.value

    def iters(self):
        a, b = C.c_int(), C.c_int()
        _check(lib().sphx_iters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get(self, field):
        nbytes = C.c_size_t()
        _check(lib().sphx_field_bytes(self._h, field, C.byref(nbytes)))
        dt = np.int32 if field in _INT_FIELDS else np.float32
        out = np.empty(nbytes.value // 4, dt)
        _check(lib().sphx_get(self._h, field, out.ctypes.data, nbytes.value))
        return out.reshape(-1, 3) if field in _VEC_FIELDS else out

    def set(self, field, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().sphx_set(self._h, field, arr.ctypes.data, arr.nbytes))

    def set_count(self, n):
        _check(lib().sphx_set_count(self._h, n))

    def run_phase(self, phase):
        _check(lib().sphx_run_phase(self._h, phase))

    def run_phase_reduce(self, phase, lo, hi):
        _check(lib().sphx_run_phase_reduce(self._h, phase, lo, hi))

    def error_total_fixed(self):
        v = C.c_longlong()
        _check(lib().sphx_error_total_fixed(self._h, C.byref(v)))
        return v.value

    def row_stats(self):
        """(total accepted pairs, longest row, histogram[128] of row lengths) of the last row build"""
        tot, mx = C.c_longlong(), C.c_int()
        hist = np.zeros(128, np.int32)
        _check(lib().sphx_row_stats(self._h, C.byref(tot), C.byref(mx), hist.ctypes.data))
        return tot.value, mx.value, hist

    def row_walk_stats(self, cut=48):
        """what ragged rows cost the quad walk (sphx_row_walk_stats): dict of wave / chunk-step counts for the last row build"""
        out = (C.c_longlong * 6)()
        lib().sphx_row_walk_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
        _check(lib().sphx_row_walk_stats(self._h, cut, out))
        keys = ("waves", "steps_walked", "steps_even_rows", "steps_cut", "steps_tail_launch", "particles_over_cut")
        return dict(zip(keys, (int(v) for v in out)), cut=cut)

    def rows_stale(self):
        v = C.c_int()
        lib().sphx_rows_stale.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _check(lib().sphx_rows_stale(self._h, C.byref(v)))
        return v.value

    def rows_partial(self):
        """PBD skin rows: launches so far that rebuilt only the rows of particles which had changed their cell"""
        v = C.c_int()
        lib().sphx_rows_partial.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _check(lib().sphx_rows_partial(self._h, C.byref(v)))
        return v.value

    def persistent_stats(self):
        """(in use, row builds, steps) of the persistent-rows mode (reserved[3] = 2)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().sphx_persistent_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _check(lib().sphx_persistent_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return bool(a.value), b.value, c.value

    def invalidate_order(self):
        """after writing API arrays through raw device pointers of a persistent-rows system"""
        _check(lib().sphx_invalidate_order(self._h))

    def device_ptr(self, field):
        p = C.c_void_p()
        _check(lib().sphx_device_ptr(self._h, field, C.byref(p)))
        return p.value

    def profile_step(self, cap=64):
        names = (C.c_char * 48 * cap)()
        ms = (C.c_float * cap)()
        cnt = C.c_int()
        _check(lib().sphx_profile_step(self._h, cap, names, ms, C.byref(cnt)))
        return [(names[i].value.decode(), ms[i]) for i in range(cnt.value)]

    def close(self):
        if getattr(self, "_h", None):
            lib().sphx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------ obstacle samplers
def _sample(fn, *args):
    n = C.c_int()
    _check(fn(*args, None, 0, C.byref(n)))
    out = np.empty((n.value, 3), np.float32)
    _check(fn(*args, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
    return out


def sample_box(lo, hi, spacing):
    """boundary particles on the surface of the axis-aligned box [lo, hi] (sphx_sample_box)"""
    f = lib().sphx_sample_box
    f.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return _sample(f, (C.c_float * 3)(*lo), (C.c_float * 3)(*hi), C.c_float(spacing))


def sample_sphere(center, radius, spacing):
    f = lib().sphx_sample_sphere
    f.argtypes = [C.c_float * 3, C.c_float, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return _sample(f, (C.c_float * 3)(*center), C.c_float(radius), C.c_float(spacing))


def sample_triangles(tri, spacing):
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    f = lib().sphx_sample_triangles
    f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return _sample(f, C.c_void_p(tri.ctypes.data), len(tri), C.c_float(spacing))


# ------------------------------------------------------------------------------------ native slab layer
SLAB_NO_OVERLAP, SLAB_SWEEP_GHOSTS = 1, 2
SLAB_EXPORTS = ["sphx_slab_rccl_unique_id", "sphx_slab_create", "sphx_slab_destroy", "sphx_slab_step", "sphx_slab_info",
                "sphx_slab_gather", "sphx_slab_iters", "sphx_slab_system", "sphx_slab_wait_seconds", "sphx_slab_set_rebalance",
                "sphx_slab_plan_cuts", "sphx_slab_plan_capacity", "sphx_slab_cut_rule", "sphx_slab_comm_info"]


def device_pci_id(ordinal):
    """PCI bus id of HIP device `ordinal` (hipDeviceGetPCIBusId), e.g. '0000:05:00.0'"""
    buf = C.create_string_buffer(64)
    lib().sphx_device_pci_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    _check(lib().sphx_device_pci_id(int(ordinal), buf, 64))
    return buf.value.decode()


def slab_plan_cuts(params, fluid, world):
    """(cuts[world + 1], counts[world]) of the initial x-slab decomposition (host-only, no GPU needed)"""
    fluid = np.ascontiguousarray(fluid, np.float32).reshape(-1, 3)
    cuts = (C.c_int * (world + 1))(); counts = (C.c_longlong * world)()
    f = lib().sphx_slab_plan_cuts
    f.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _check(f(C.byref(params), fluid.ctypes.data, len(fluid), world, cuts, counts))
    return list(cuts), list(counts)


def slab_plan_capacity(params, fluid, world):
    """particle capacity of every slab's engine for this scene and slab count (host-only, no GPU needed)"""
    fluid = np.ascontiguousarray(fluid, np.float32).reshape(-1, 3)
    cap = C.c_longlong()
    f = lib().sphx_slab_plan_capacity
    f.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    _check(f(C.byref(params), fluid.ctypes.data, len(fluid), world, C.byref(cap)))
    return cap.value


def slab_cut_rule(owned_left, owned_right, width_left, width_right, ghost=1, tolerance=0.05):
    f = lib().sphx_slab_cut_rule
    f.argtypes = [C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float]
    return f(owned_left, owned_right, width_left, width_right, ghost, tolerance)


def rccl_unique_id():
    """128-byte RCCL bootstrap token (rank 0 creates it; hand it to every rank over any side channel)"""
    buf = C.create_string_buffer(128)
    _check(lib().sphx_slab_rccl_unique_id(buf))
    return buf.raw


class SlabGroup:
    """include/sphx_slab.h: the x-slabs of one simulation driven by this process (one with RCCL, all with loopback)"""

    def __init__(self, params, fluid, boundary, world, first_rank=0, local_ranks=None, rccl_id=None, flags=0, velocity=None):
        L = lib()
        L.sphx_slab_create.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.sphx_slab_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.sphx_slab_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4
        L.sphx_slab_gather.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(C.c_int)]
        L.sphx_slab_iters.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sphx_slab_system.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.sphx_slab_wait_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.sphx_slab_destroy.argtypes = [C.c_void_p]
        L.sphx_slab_set_rebalance.argtypes = [C.c_void_p, C.c_int, C.c_float]
        fluid = np.ascontiguousarray(fluid, np.float32).reshape(-1, 3)
        boundary = np.ascontiguousarray(boundary, np.float32).reshape(-1, 3)
        vel = None if velocity is None else np.ascontiguousarray(velocity, np.float32).reshape(-1, 3)
        self.world = world
        self.local = world if local_ranks is None else local_ranks
        self.n = len(fluid)
        h = C.c_void_p()
        _check(L.sphx_slab_create(C.byref(params), fluid.ctypes.data, None if vel is None else vel.ctypes.data, len(fluid),
                                  boundary.ctypes.data, len(boundary), world, first_rank, self.local, rccl_id, flags, C.byref(h)))
        self._h = h

    def step(self, n=1):
        ms = C.c_float()
        _check(lib().sphx_slab_step(self._h, n, C.byref(ms)))
        return ms.value

    def set_rebalance(self, every_steps, tolerance=0.05):
        _check(lib().sphx_slab_set_rebalance(self._h, every_steps, tolerance))

    def info(self, index=0):
        v = [C.c_int() for _ in range(4)]
        _check(lib().sphx_slab_info(self._h, index, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)          # x0, x1, owned, held

    def gather(self, index=0):
        cap = self.info(index)[2]
        ids = np.empty(cap, np.int32); pos = np.empty((cap, 3), np.float32); vel = np.empty((cap, 3), np.float32)
        den = np.empty(cap, np.float32)
        cnt = C.c_int()
        _check(lib().sphx_slab_gather(self._h, index, cap, ids.ctypes.data, pos.ctypes.data, vel.ctypes.data, den.ctypes.data, C.byref(cnt)))
        return ids, pos, vel, den

    def gather_all(self):
        """owned state of every local slab, concatenated and ordered by original particle index"""
        parts = [self.gather(i) for i in range(self.local)]
        ids = np.concatenate([p[0] for p in parts])
        order = np.argsort(ids, kind="stable")
        return (ids[order],) + tuple(np.concatenate([p[k] for p in parts])[order] for k in (1, 2, 3))

    def iters(self):
        a, b = C.c_int(), C.c_int()
        _check(lib().sphx_slab_iters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def row_capacity(self, index=0):
        """entries per neighbour row of local slab `index`'s engine (sphx_slab_system + sphx_row_capacity)"""
        sys_h, cap = C.c_void_p(), C.c_int()
        _check(lib().sphx_slab_system(self._h, index, C.byref(sys_h)))
        lib().sphx_row_capacity.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _check(lib().sphx_row_capacity(sys_h, C.byref(cap)))
        return cap.value

    def comm_info(self):
        """the transport as it reports itself: kind, ranks and own rank of the communicator (ncclCommCount / ncclCommUserRank),
        payload bytes sent / received, exchanges and all-reduces posted by this process"""
        kind, ranks, rank = C.c_int(), C.c_int(), C.c_int()
        ctr = (C.c_longlong * 4)()
        lib().sphx_slab_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
        _check(lib().sphx_slab_comm_info(self._h, C.byref(kind), C.byref(ranks), C.byref(rank), ctr))
        return {"transport": "rccl" if kind.value == 1 else "loopback", "ranks": ranks.value, "rank": rank.value,
                "bytes_sent": int(ctr[0]), "bytes_received": int(ctr[1]), "exchanges": int(ctr[2]), "allreduces": int(ctr[3])}

    def wait_seconds(self):
        v = C.c_double()
        _check(lib().sphx_slab_wait_seconds(self._h, C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            lib().sphx_slab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def row_capacity(system):
    cap = C.c_int()
    _check(lib().sphx_row_capacity(system._h, C.byref(cap)))
    return cap.value


def use_stream(hip_stream_handle):
    """enqueue all engine work on a caller-owned hipStream_t (int handle), e.g. torch's current stream"""
    _check(lib().sphx_use_stream(C.c_void_p(hip_stream_handle)))


def cell_columns(device_xyz_ptr, n, cell_length, device_out_ptr):
    _check(lib().sphx_cell_columns(C.c_void_p(device_xyz_ptr), n, cell_length, C.c_void_p(device_out_ptr)))


def fastmath_selftest(radius, samples=1 << 28):
    bad = (C.c_uint * 3)(); en = (C.c_int * 2)()
    lib().sphx_fastmath_selftest.argtypes = [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]
    _check(lib().sphx_fastmath_selftest(radius, samples, bad, en))
    return list(bad), list(en)


def sync():
    _check(lib().sphx_sync())


def kernel_timer(enable, name_filter=""):
    _check(lib().sphx_kernel_timer(int(enable), name_filter.encode()))


def kernel_timer_collect(cap=64):
    """{span name: (total_ms, launches)} since kernel_timer(True, ...)"""
    names = (C.c_char * 48 * cap)()
    ms = (C.c_float * cap)()
    cnt = (C.c_int * cap)()
    k = C.c_int()
    _check(lib().sphx_kernel_timer_collect(cap, names, ms, cnt, C.byref(k)))
    return {names[i].value.decode(): (ms[i], cnt[i]) for i in range(k.value)}


def eval_kernels(r3, radius):
    r3 = np.ascontiguousarray(r3, np.float32).reshape(-1, 3)
    n = len(r3)
    W = np.empty(n, np.float32); G = np.empty((n, 3), np.float32)
    V = np.empty(n, np.float32); S = np.empty((n, 3), np.float32)
    _check(lib().sphx_eval_kernels(r3.ctypes.data, n, radius, W.ctypes.data, G.ctypes.data, V.ctypes.data,
                                   S.ctypes.data))
    return W, G, V, S


def ieee_probe(a, b, c):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    n = len(a)
    q = np.empty(n, np.float32); r = np.empty(n, np.float32)
    t = np.empty(n, np.int32); m = np.empty(n, np.float32)
    _check(lib().sphx_ieee_probe(a.ctypes.data, b.ctypes.data, c.ctypes.data, n, q.ctypes.data, r.ctypes.data,
                                 t.ctypes.data, m.ctypes.data))
    return q, r, t, m


# ------------------------------------------------------------------------------------ snapshots
# sphx_snapshot_save / sphx_snapshot_load (include/sphx_c.h): everything a run needs to continue
# bit-identically, in the CURRENT array order; the loader restores that order without re-sorting.
class _Loaded(System):
    """a System adopted from sphx_snapshot_load"""

    def __init__(self, handle):
        self._h = handle
        n, nb, cells = C.c_int(), C.c_int(), C.c_int()
        _check(lib().sphx_counts(self._h, C.byref(n), C.byref(nb), C.byref(cells)))
        self.n, self.nb, self.cells = n.value, nb.value, cells.value
        P = Params()
        _check(lib().sphx_get_params(self._h, C.byref(P)))
        self.params = P


def save_snapshot(system, path):
    _check(lib().sphx_snapshot_save(system._h, os.fsencode(path)))


def load_snapshot(path):
    """returns a System that continues the saved run (no constructor step, no re-sort)"""
    h = C.c_void_p()
    _check(lib().sphx_snapshot_load(os.fsencode(path), C.byref(h)))
    return _Loaded(h)
